// so3x_protnet.hpp -- shapes, parameter layout and buffer plans of the ProtNet docking denoiser (reference models.py:212-319),
// shared by the exact-fp32 form (so3x_protnet.hip) and the bf16 matrix-core form of the class-default width (so3x_protnet_bf16.hip).
#pragma once
#include "so3x_planenet.hpp"

namespace so3x {
namespace prot {

using plane::Carve;
using plane::LayerOff;

constexpr int RES = 21;   // prot_util.RES_COUNT: 20 residue types + "---"

// One batch: B complexes = 2 B chains ("sequences": s < B the receptor of complex s, s >= B the ligand of complex s - B), every
// chain padded to Lp rows inside the kernels' own buffers (the reference pads with pad_sequence, models.py:277-298).
struct Dims {
  int64_t B, Lp;
  int d, H, T, Cd, F;   // width, heads, encoder layers (t_depth), conv layers (c_depth), feed-forward width (torch's default 2048)
  int64_t S() const { return 2 * B; }
  int64_t N() const { return 2 * B * Lp; }          // padded tokens
  int64_t R() const { return 2 * B * (Lp + 2); }    // rows of a halo buffer: one zero row before and after every chain
  int pd() const { return d / 2; }                  // pos_emb width (models.py:217)
  int ad() const { return d / 4; }                  // ang_emb width
  int rd() const { return d - d / 2 - d / 4; }      // res_conv output width
  int dh() const { return d / H; }
  int pw() const { return 3 * d + 6; }              // [time | rec pool | rec pos | lig pool | lig pos] (models.py:311)
};
inline bool dims_ok(const Dims& s) {
  return s.B >= 0 && s.B <= 16383 && s.Lp >= 1 && s.Lp <= 4096 && s.d >= 8 && s.d <= 1024 && s.d % 4 == 0 && s.H >= 1 && s.d % s.H == 0 &&
         s.T >= 1 && s.T <= 64 && s.Cd >= 2 && s.Cd <= 34 && s.F >= 1 && s.F <= 16384 && s.S() * s.H <= 65535 && s.R() * (int64_t)(s.F > 3 * s.d ? s.F : 3 * s.d) < (int64_t(1) << 40);
}

// Offsets (floats) into the flat parameter buffer = state_dict order of models.ProtNet (models.py:213-273): pos_emb.{positional,
// post_scale}.{weight, bias}, ang_emb.(same), res_conv.{0, 2 .. c_depth - 1 (.layer.0), c_depth}.{weight [out][in][3], bias},
// lig_tf.encoder.layers.{l}.(the 12 tensors of nn.TransformerEncoderLayer), lig_tf.encoder.norm.{weight, bias},
// lig_emb_pool.{pool.0.weight, pool.0.bias, lin.weight, lin.bias}, lig_pos_pool.pool.0.{weight, bias}, rec_tf.(same), rec_emb_pool,
// rec_pos_pool, last.0.0.{weight [d][3 d + 6], bias}, last.{1, 2, 3}.layer.0.{weight, bias}, last.4.{weight [6][d], bias}.
struct Tf { int64_t base, per_layer, gF, bF; };   // an encoder: its layers, then the final norm
struct Pool { int64_t wpool, bpool, wlin, blin, wppool, bppool; };   // PoolRN then PoolPos
struct POff {
  int64_t wpp, bpp, wpps, bpps, wap, bap, waps, baps;
  int64_t cw[34], cb[34];
  int cin[34], cout[34];
  Tf lig_tf, rec_tf;
  Pool lig, rec;
  int64_t w0, b0, wr[3], br[3], wout, bout, total;
  int d, F;
  __host__ __device__ LayerOff layer(const Tf& tf, int l) const {
    plane::ParamOff o;
    o.d = d;
    o.F = F;
    o.per_layer = tf.per_layer;
    LayerOff lo = o.layer(l);
    int64_t* f = &lo.wqkv;
    for (int i = 0; i < 12; i++) f[i] += tf.base;
    return lo;
  }
};
inline POff param_offsets(const Dims& s) {
  POff o;
  o.d = s.d;
  o.F = s.F;
  const int64_t d = s.d, F = s.F, pd = s.pd(), ad = s.ad();
  int64_t p = 0;
  o.wpp = p; p += pd * 3;
  o.bpp = p; p += pd;
  o.wpps = p; p += pd * pd;
  o.bpps = p; p += pd;
  o.wap = p; p += ad * 9;
  o.bap = p; p += ad;
  o.waps = p; p += ad * ad;
  o.baps = p; p += ad;
  for (int i = 0; i < s.Cd; i++) {
    o.cin[i] = i == 0 ? RES : s.d;
    o.cout[i] = i == s.Cd - 1 ? s.rd() : s.d;
    o.cw[i] = p; p += (int64_t)o.cout[i] * o.cin[i] * 3;
    o.cb[i] = p; p += o.cout[i];
  }
  const int64_t per_layer = 4 * d * d + 2 * d * F + 9 * d + F;
  auto tf = [&](Tf& t) {
    t.base = p;
    t.per_layer = per_layer;
    p += per_layer * s.T;
    t.gF = p; p += d;
    t.bF = p; p += d;
  };
  auto pool = [&](Pool& q) {
    q.wpool = p; p += d;
    q.bpool = p; p += 1;
    q.wlin = p; p += d * d;
    q.blin = p; p += d;
    q.wppool = p; p += d;
    q.bppool = p; p += 1;
  };
  tf(o.lig_tf);
  pool(o.lig);
  tf(o.rec_tf);
  pool(o.rec);
  o.w0 = p; p += d * s.pw();
  o.b0 = p; p += d;
  for (int i = 0; i < 3; i++) {
    o.wr[i] = p; p += d * d;
    o.br[i] = p; p += d;
  }
  o.wout = p; p += 6 * d;
  o.bout = p; p += 6;
  o.total = p;
  return o;
}

}  // namespace prot
}  // namespace so3x
