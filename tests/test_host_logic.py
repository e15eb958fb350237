"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol that
include/so3x.h declares, host-side entry points match the goldens, the Python host
layer keeps the reference's interface, refuses CPU tensors loudly, and the
data-parallel plumbing works over gloo with world_size 2."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, PKG

from so3x import backend as B


def test_library_exports_every_header_symbol():
    hdr = open(os.path.join(ROOT, "include", "so3x.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(so3x_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 25
    lib = C.CDLL(B.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/so3x.h but not exported"
    assert declared == set(B.SYMBOLS)
    assert B.lib().so3x_abi_version() == 8


def test_no_oracle_or_cpu_fallback_in_product():
    # the product package must not import or reference the oracle (test infrastructure only)
    for dirpath, _, files in os.walk(PKG):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.lower(), f"{f} mentions the oracle"


def test_product_library_has_no_environment_switches():
    """libso3x.so reads no environment variable (getenv races with setenv, and an exported variable must never change
    arithmetic) and ships ONE bf16 form of the chain kernel (plus its f16-operand leg, selected by the precision argument); the A/B forms live in libso3x_ab.so (-DSO3X_AB_BUILD), which
    only tools/ab and one parity test load by path."""
    dyn = subprocess.run(["nm", "-D", "--undefined-only", B.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in dyn
    syms = subprocess.run(["nm", B.LIB_PATH], capture_output=True, text=True, check=True).stdout  # mangled: I<PREC>E... = template args
    stubs = [l for l in syms.splitlines() if "__device_stub__k_p_sample_chainI" in l]
    # <fp32 parity form>, <bf16 product form>, and (round 4) the product form's f16-operand leg -- a precision argument of the C ABI
    # (SO3X_PREC_F16), not an environment switch
    assert len(stubs) == 3, stubs
    assert sum("k_p_sample_chainILi1ELb1ELb1ELb1ELb0E" in l for l in stubs) == 1 and sum("k_p_sample_chainILi1ELb1ELb1ELb1ELb1E" in l for l in stubs) == 1
    assert not any("k_train_fwdI" in l for l in syms.splitlines())   # the fused noising + forward experiment is A/B-only too
    ab = os.path.join(os.path.dirname(B.LIB_PATH), "libso3x_ab.so")
    assert os.path.exists(ab), "make -C csrc builds the A/B library beside the product one"
    assert "getenv" in subprocess.run(["nm", "-D", "--undefined-only", ab], capture_output=True, text=True, check=True).stdout
    for f in os.listdir(os.path.join(PKG, "so3x")):   # and the package never loads it
        if f.endswith(".py"):
            assert "libso3x_ab" not in open(os.path.join(PKG, "so3x", f)).read(), f


def test_schedule_matches_golden(golden):
    g = golden["schedule"]
    for T in (100, 1000):
        betas = B.cosine_beta_schedule(T)
        assert np.array_equal(betas, g[f"betas64_{T}"])
        tab = B.schedule_from_betas(betas)
        names = ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
                 "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod",
                 "sqrt_recipm1_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
                 "posterior_mean_coef1", "posterior_mean_coef2")
        for i, nme in enumerate(names):
            assert np.allclose(tab[i], g[f"{nme}_{T}"], rtol=2e-7, atol=0), nme
        assert np.allclose(tab[12], np.exp(0.5 * tab[9].astype(np.float64)), rtol=2e-7)
    assert abs(float(tab[6][999]) - 20291) < 1 and tab[10][0] == 1 and tab[11][0] == 0


def test_knots_and_freqs_match_fixture(golden):
    k, w = B.igso3_knots()
    g = golden["igso3_knots"]
    assert np.array_equal(k, g["knots"]) and np.array_equal(w, g["haar_w"])
    import math
    fr = B.posemb_freqs(28)
    ref = torch.exp(torch.arange(28) * -(math.log(10000) / 27)).numpy()
    assert np.array_equal(fr, ref)


def test_error_paths_return_codes_not_crashes():
    lib = B.lib()
    assert lib.so3x_so3_scale(None, None, None, C.c_int64(1), None, C.c_int64(4)) == -1
    assert lib.so3x_quat_to_rmat(None, None, None, C.c_int64(0)) == 0
    assert lib.so3x_p_sample_chain(None, None, None, C.c_int(10), None, None, None, None, C.c_int(3), C.c_int(5), None, None,
                                   C.c_uint64(0), C.c_uint64(0), C.c_int64(0), C.c_int64(8), C.c_int(1), None,
                                   C.c_size_t(0)) == -1  # t_start - n_steps + 1 < 0
    assert b"workspace" in lib.so3x_error_string(-2)
    ws = lib.so3x_p_sample_workspace_bytes(C.c_int(1000), C.c_int(1))
    # bf16: weight image + [T][96] effective-bias table + [T][3 KB] per-timestep layer-0 fragments
    # weight image + per-timestep bias rows and layer-0 fragments + the 4,608-byte CDF records the chain kernel stages in LDS
    assert 53 * 1024 + 1000 * (96 * 4 + 3072 + 4608) <= ws <= 60 * 1024 + 1000 * (96 * 4 + 3072 + 4608)
    ws32 = lib.so3x_p_sample_workspace_bytes(C.c_int(1000), C.c_int(0))
    assert 1000 * 96 * 4 < ws32 < 200 * 1024 + 1000 * 96 * 4


def test_error_paths_of_the_round_1_additions():
    """argument validation returns codes before anything touches a device (no GPU needed)"""
    lib = B.lib()
    i64, ci = C.c_int64, C.c_int
    assert lib.so3x_six2rmat(None, None, None, i64(-1)) == -1 and lib.so3x_six2rmat(None, None, None, i64(0)) == 0
    assert lib.so3x_six2rmat(None, None, None, i64(5)) == -1                      # null buffers with n > 0
    assert lib.so3x_six2rmat_bwd(None, None, None, None, i64(0)) == 0 and lib.so3x_six2rmat_bwd(None, None, None, None, i64(3)) == -1
    assert lib.so3x_log_rmat_bwd(None, None, None, None, i64(-2)) == -1
    assert lib.so3x_rmat_dist_bwd(None, None, None, None, None, None, i64(4)) == -1
    assert lib.so3x_prevstep_loss(None, None, ci(10), None, None, None, None, i64(1), i64(0), None, None, None, None, C.c_size_t(0)) == -1
    assert lib.so3x_prevstep_loss6(None, None, ci(10), None, None, None, None, i64(1), i64(4), None, None, None, C.c_size_t(0)) == -1
    lib.so3x_prevstep_workspace_bytes.restype = C.c_size_t
    assert 0 < lib.so3x_prevstep_workspace_bytes(i64(1 << 20)) <= 1 << 20
    # n_out must be 3 or 6 on the six network entry points
    for bad in (0, 4, 7):
        assert lib.so3x_mlp_fwd(None, None, None, None, i64(1), None, i64(0), ci(bad), ci(1), ci(10), None, C.c_size_t(0)) == -1
        assert lib.so3x_resnet_fwd(None, None, None, None, i64(1), None, i64(0), ci(bad), ci(1), ci(10), None, C.c_size_t(0)) == -1
        assert lib.so3x_mlp_bwd(None, None, None, None, i64(1), None, None, i64(0), ci(bad), ci(1), ci(10), None, None, C.c_size_t(0)) == -1
    # workspace too small -> -2 (before any launch)
    one = C.c_void_p(1)  # never dereferenced: the size check comes first
    assert lib.so3x_mlp_fwd(None, one, one, one, i64(1), one, i64(8), ci(3), ci(1), ci(10), one, C.c_size_t(16)) == -2
    assert lib.so3x_resnet_fwd(None, one, one, one, i64(1), one, i64(8), ci(3), ci(1), ci(10), one, C.c_size_t(16)) == -2
    assert b"argument" in lib.so3x_error_string(-1).lower() or b"invalid" in lib.so3x_error_string(-1).lower()


def test_cpu_tensors_are_refused_loudly():
    from so3x import util
    with pytest.raises(B.So3xError, match="no CPU path"):
        util.quat_to_rmat(torch.randn(4, 4))
    with pytest.raises(B.So3xError):
        util.so3_scale(torch.eye(3)[None], torch.ones(1))


def test_reference_interface_is_kept(golden):
    from so3x.diffusion import SO3Diffusion
    from so3x.so3_train import RotPredict
    from so3x import util, distributions
    net = RotPredict(out_type="skewvec")
    keys = list(net.state_dict().keys())
    assert keys == [f"net.{l}.{k}" for l in (0, 2, 4, 6, 8) for k in ("weight", "bias")]
    assert sum(p.numel() for p in net.parameters()) == 17358
    assert net.flat_params().numel() == 17358
    # checkpoints interchange with the reference's: load the golden (reference-initialised) weights
    g = golden["score_mlp"]
    net.load_state_dict({f"net.{l}.{k}": torch.from_numpy(g[f"net_{l}_{k}"]) for l in (0, 2, 4, 6, 8) for k in ("weight", "bias")})
    proc = SO3Diffusion(net, timesteps=100, loss_type="skewvec")
    assert proc.num_timesteps == 100 and proc.identity.shape == (3, 3)
    gs = golden["schedule"]
    for nme in ("betas", "sqrt_alphas_cumprod", "posterior_mean_coef1", "posterior_log_variance_clipped"):
        assert np.allclose(getattr(proc, nme).numpy(), gs[f"{nme}_100"], rtol=2e-7, atol=0)
    for m in ("forward", "p_losses", "q_sample", "p_sample", "p_sample_loop", "predict_start_from_noise", "q_posterior",
              "p_mean_variance", "q_mean_variance"):
        assert callable(getattr(proc, m))
    for f in ("quat_to_rmat", "rmat_dist", "so3_lerp", "so3_scale", "log_rmat", "skew2vec", "vec2skew", "aa_to_rmat",
              "rmat_to_aa"):
        assert callable(getattr(util, f))
    v = torch.randn(5, 3)
    assert torch.equal(util.skew2vec(util.vec2skew(v)), v)
    S = util.vec2skew(v)
    assert torch.equal(S, -S.transpose(-1, -2)) and torch.equal(S[:, 2, 1], v[:, 0]) and torch.equal(S[:, 2, 0], -v[:, 1])
    explicit = SO3Diffusion(net, betas=torch.linspace(1e-4, 0.02, 50))
    assert explicit.num_timesteps == 50
    rot = RotPredict()                      # the reference's default head: out_type="rotmat", Linear(65, 6) + six2rmat
    assert rot.out_type == "rotmat" and rot.net[8].out_features == 6 and rot.flat_params().numel() == 17556
    with pytest.raises(ValueError):
        RotPredict(out_type="quat")
    assert SO3Diffusion(rot, timesteps=10, loss_type="prevstep").loss_type == "prevstep"
    with pytest.raises(ValueError):
        SO3Diffusion(net, timesteps=10, loss_type="l2")
    for f in ("six2rmat", "rmat2six"):
        assert callable(getattr(util, f))
    assert torch.equal(util.rmat2six(torch.arange(18.).reshape(2, 3, 3)), torch.tensor([[0., 1, 2, 3, 4, 5], [9, 10, 11, 12, 13, 14]]))
    assert hasattr(distributions.IsotropicGaussianSO3, "sample") and hasattr(distributions.IsotropicGaussianSO3, "log_prob")


def test_shard_ranges_cover_batch_exactly():
    from so3x.parallel import shard_range
    for n in (0, 1, 7, 64, 2 ** 20, 2 ** 22 + 3):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


_WORKER = r'''
import os, sys, torch
sys.path.insert(0, sys.argv[1])
from so3x import parallel
from so3x.so3_train import RotPredict
ctx = parallel.init(backend="gloo", device="cpu")
assert ctx.world_size == 2
torch.manual_seed(ctx.rank)            # different init per rank on purpose
net = RotPredict(out_type="skewvec")
parallel.broadcast_parameters(net, ctx)
flat = torch.cat([p.data.reshape(-1) for p in net.parameters()])
ref = [torch.zeros_like(flat) for _ in range(2)]
torch.distributed.all_gather(ref, flat)
assert torch.equal(ref[0], ref[1]), "parameters differ after broadcast"
for i, p in enumerate(net.parameters()):
    p.grad = torch.full_like(p, float(ctx.rank + 1) * (i + 1))
parallel.allreduce_gradients(net, ctx)
for i, p in enumerate(net.parameters()):
    assert torch.allclose(p.grad, torch.full_like(p, 1.5 * (i + 1))), "gradient mean wrong"
# the flat path: the score network's gradient as ONE tensor (what the fused backward returns), .grad = views of it
net.zero_grad(set_to_none=True)
flat_g = torch.full((17358,), float(ctx.rank + 1))
net._install_flat_grad(flat_g)
assert net.flat_grad() is flat_g
class _Opt:  # an optimizer that folds the 1/world into its update (so3x.optim.Adam does)
    grad_scale = 1.0
opt = _Opt()
parallel.allreduce_gradients(net, ctx, n_local=5, n_global=10, optimizer=opt)
assert net.flat_grad() is flat_g, "the all-reduce must run in place on the flat gradient"
assert torch.allclose(flat_g * opt.grad_scale, torch.full_like(flat_g, 1.5)) and opt.grad_scale == 0.5
assert all(p.grad.data_ptr() >= flat_g.data_ptr() for p in net.parameters())
# unequal shards: weighted by n_local / n_global (3 + 7 samples)
g2 = torch.full((8,), float(ctx.rank + 1))
parallel.allreduce_flat(g2, ctx, n_local=3 if ctx.rank == 0 else 7, n_global=10)
assert torch.allclose(g2, torch.full_like(g2, 0.3 * 1 + 0.7 * 2))
# a decision every rank must take together (so3_lock_train's skip-on-NaN)
assert parallel.any_rank_true(torch.tensor(ctx.rank == 1), ctx) is True
assert parallel.any_rank_true(torch.tensor(False), ctx) is False
m = parallel.mean_scalar(torch.tensor(float(ctx.rank)), ctx)
assert abs(m - 0.5) < 1e-7
lo, hi = parallel.shard_range(10, ctx.rank, ctx.world_size)
assert (lo, hi) == ((0, 5) if ctx.rank == 0 else (5, 10))
parallel.finalize(ctx)
print("OK", ctx.rank)
'''


def test_gloo_world_size_2_gradient_allreduce(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2")
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(script), PKG], env=e, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=180)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
        assert "OK" in o


def test_flat_name_shims_resolve_like_the_reference_layout(tmp_path):
    """`from diffusion import SO3Diffusion` etc. (the reference's flat module names) resolve through compat/."""
    code = ("from diffusion import SO3Diffusion, SE3Diffusion\n"
            "from distributions import IsotropicGaussianSO3, IGSO3xR3\n"
            "from so3_train import RotPredict\n"
            "from models import SinusoidalPosEmb\n"
            "from util import *\n"
            "assert callable(quat_to_rmat) and callable(so3_scale) and callable(MMD) and AffineT is not None\n"
            "p = SO3Diffusion(RotPredict(out_type='skewvec'), timesteps=10)\n"
            "print('OK', p.num_timesteps)\n")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(PKG, "compat"), PKG]))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=str(tmp_path), timeout=180)
    assert r.returncode == 0 and "OK 10" in r.stdout, r.stderr


def test_small_helpers_of_the_reference_namespace():
    """util.{identity, masked_mean, init_from_dict, to_device}, diffusion.noise_like, models.Siren (reference util.py:426-475,
    diffusion.py:19-22, models.py:37-72): host-side helpers, plain torch"""
    from so3x import util, models
    from so3x.diffusion import noise_like
    assert util.identity(3) == 3
    x = torch.arange(12.).reshape(3, 4)
    m = torch.tensor([[True, True, False, False], [False] * 4, [True] * 4])
    assert torch.equal(util.masked_mean(x.clone(), m), torch.tensor([0.5, 0.0, 9.5]))
    lin, act = util.init_from_dict({"in_features": 4, "out_features": 2, "negative_slope": 0.3, "unused": 1},
                                   torch.nn.Linear, torch.nn.LeakyReLU)
    assert lin.weight.shape == (2, 4) and act.negative_slope == 0.3
    a, (b, c) = util.to_device("cpu", torch.ones(2), (torch.zeros(1), torch.zeros(3)))
    assert a.shape == (2,) and b.shape == (1,) and c.shape == (3,)
    n = noise_like((5, 3, 3), "cpu", repeat=True)
    assert n.shape == (5, 3, 3) and torch.equal(n[0], n[4]) and not torch.equal(noise_like((5, 3), "cpu")[0], noise_like((5, 3), "cpu")[1])
    s = models.Siren(3, 8, scale=30)
    assert s(torch.randn(7, 3)).shape == (7, 8) and float(s.positional.weight.detach().abs().max()) <= 30 * (6 / 3) ** 0.5 + 1e-6
    assert all(not p.requires_grad for p in models.Siren(3, 8, optimize=False, post_scale=False).parameters())


def test_score_network_parameters_live_in_one_flat_buffer(golden):
    """so3x.flat: the nn.Linear parameters are views of one flat tensor in state_dict order, and stay so through
    load_state_dict / in-place updates / deepcopy; the flat gradient comes back as .grad views without copies"""
    import copy
    from so3x.so3_train import RotPredict
    from so3x.so3_lock_train import RotPredict as Wide
    for cls, n in ((RotPredict, 17358), (Wide, 392448)):
        torch.manual_seed(0)
        net = cls(out_type="skewvec")
        flat = net.flat_data()
        assert flat.numel() == n and net._flat_ok()
        assert torch.equal(flat, torch.cat([p.detach().reshape(-1) for p in net.state_dict().values()]))
        with torch.no_grad():
            next(net.net.parameters()).add_(1.0)                  # an optimizer's in-place update is seen at once
        assert torch.equal(net.flat_data(), torch.cat([p.detach().reshape(-1) for p in net.parameters()]))
        sd = {k: torch.randn_like(v) for k, v in net.state_dict().items()}
        net.load_state_dict(sd)
        assert net._flat_ok() and torch.equal(net.flat_data(), torch.cat([v.reshape(-1) for v in sd.values()]))
        twin = copy.deepcopy(net)
        assert torch.equal(twin.flat_data(), net.flat_data()) and twin.flat_data().data_ptr() != net.flat_data().data_ptr()
        with torch.no_grad():
            twin.flat_data().zero_()
        assert float(next(twin.net.parameters()).detach().abs().sum()) == 0.0 and float(net.flat_data().abs().sum()) > 0
        net = net.double().float()                                   # _apply re-homes the parameters: adopted again
        assert net._flat_ok()
        # autograd: the flat gradient is installed as views, and accumulates when asked to
        fp = net.flat_params()
        w = torch.arange(n, dtype=torch.float32)
        (fp * w).sum().backward()
        g = net.flat_grad()
        assert g is not None and torch.equal(g, w)
        assert all(p.grad.data_ptr() == g.data_ptr() + 4 * off for p, off in
                   zip(net.net.parameters(), np.cumsum([0] + [q.numel() for q in net.net.parameters()])[:-1]))
        (net.flat_params() * w).sum().backward()                     # second backward without zero_grad: accumulate
        assert torch.equal(torch.cat([p.grad.reshape(-1) for p in net.net.parameters()]), 2 * w)
        # torch.autograd.grad gets the gradients and leaves every .grad alone; backward(inputs=...) fills the named ones only
        net.zero_grad(set_to_none=True)
        ps = list(net.net.parameters())
        gs = torch.autograd.grad((net.flat_params() * w).sum(), ps)
        assert torch.equal(torch.cat([g_.reshape(-1) for g_ in gs]), w) and all(p.grad is None for p in ps)
        (net.flat_params() * w).sum().backward(inputs=[ps[1]])
        assert ps[1].grad is not None and all(p.grad is None for i, p in enumerate(ps) if i != 1)
        assert torch.equal(ps[1].grad.reshape(-1), w[ps[0].numel():ps[0].numel() + ps[1].numel()])
        assert net.flat_grad() is None and torch.equal(net.gather_flat_grad()[ps[0].numel():ps[0].numel() + ps[1].numel()], ps[1].grad.reshape(-1))


def test_flat_buffer_check_notices_replaced_parameters_and_modules():
    """so3x.flat's per-step check is ten pointer comparisons while nothing has been registered anywhere in the process (torch's
    global registration hooks); every way of re-homing a parameter must still be noticed: a new nn.Parameter on a layer, a replaced
    layer, load_state_dict(assign=True), a caller assigning p.data -- flat_data() then adopts the parameters again and is current"""
    from torch import nn
    from so3x.so3_train import RotPredict
    torch.manual_seed(0)

    def current(net):
        return torch.cat([p.detach().reshape(-1) for p in net.net.parameters()])

    net = RotPredict(out_type="skewvec")
    lin = next(m for m in net.net.modules() if isinstance(m, nn.Linear))
    assert net._flat_ok() and net._flat_ok()                          # second call: the fast path
    lin.weight = nn.Parameter(torch.randn_like(lin.weight))           # a new Parameter object on an old layer
    assert not net._flat_ok()
    assert torch.equal(net.flat_data(), current(net)) and net._flat_ok()
    idx = next(i for i, m in enumerate(net.net) if isinstance(m, nn.Linear))
    net.net[idx] = nn.Linear(net.net[idx].in_features, net.net[idx].out_features)   # a replaced layer
    assert not net._flat_ok()
    assert torch.equal(net.flat_data(), current(net)) and net._flat_ok()
    sd = {k: torch.randn_like(v) for k, v in net.state_dict().items()}
    net.load_state_dict(sd, assign=True)                               # parameters re-created from the given tensors
    assert torch.equal(net.flat_data(), torch.cat([v.reshape(-1) for v in sd.values()])) and net._flat_ok()
    p0 = next(net.net.parameters())
    p0.data = torch.zeros_like(p0)                                     # registers nothing: the pointer check catches it
    assert not net._flat_ok()
    assert torch.equal(net.flat_data(), current(net)) and float(net.flat_data()[:p0.numel()].abs().sum()) == 0.0
    other = RotPredict(out_type="skewvec")                             # registrations elsewhere only cost one full check
    assert net._flat_ok() and other._flat_ok()


def test_gradients_from_a_plain_torch_path_are_gathered():
    from so3x.so3_train import RotPredict
    net = RotPredict(out_type="skewvec")
    for i, p in enumerate(net.net.parameters()):
        p.grad = torch.full_like(p, float(i))
    assert net.flat_grad() is None
    g = net.gather_flat_grad()
    assert g.numel() == 17358 and net.flat_grad() is g and float(g[0]) == 0.0 and float(g[-1]) == 9.0


def test_torch_operator_library_registers_the_hot_path():
    """libso3x_torch.so: TORCH_LIBRARY(so3x, ...) schemas for every op of the SO(3) hot path, fake-tensor kernels so that
    shapes propagate without a GPU, and a dispatcher that refuses CPU tensors (no CPU kernels exist)"""
    ops = B.ops()
    names = ("quat_to_rmat", "log_rmat", "log_rmat_vec", "exp_skewvec", "orthogonalise", "so3_scale", "aa_to_rmat", "rmat_to_aa",
             "so3_lerp", "rmat_dist", "rmul", "igso3_eps_ft", "igso3_build_tables", "igso3_build_guide", "igso3_sample",
             "igso3_logprob_score", "mlp_fwd", "mlp_fwd_stash", "mlp_bwd", "q_sample_target", "p_mean", "p_sample_chain",
             "p_sample_chain_out", "train_fwd", "train_bwd", "adam_step")
    for n in names:
        schema = str(getattr(ops, n).default._schema)
        assert schema.startswith(f"so3x::{n}("), schema
    assert "Tensor(a!) params" in str(ops.adam_step.default._schema) and "Tensor(a!)? rng_counter" in str(ops.train_fwd.default._schema)
    with pytest.raises(NotImplementedError):
        ops.quat_to_rmat(torch.randn(3, 4))            # the dispatcher has no CPU kernel to offer
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        x = torch.empty(7, 3, 3, device="cuda")
        t = torch.empty(7, dtype=torch.int64, device="cuda")
        p = torch.empty(17358, device="cuda")
        assert ops.mlp_fwd(p, x, t, 1, 1, 100).shape == (7, 3)
        assert ops.log_rmat_vec(x).shape == (7, 3) and ops.rmat_dist(x, x).shape == (7,)
        out, zs = ops.mlp_fwd_stash(p, x, t, 1, 100)
        assert out.shape == (7, 3) and zs.dtype == torch.uint8 and zs.numel() == 17 * 1024
        assert ops.mlp_bwd(p, x, t, 1, out, 1, 100, zs).shape == (17358,)


def test_bench_gpus_n_starts_its_own_ranks_without_touching_a_gpu():
    """VERDICT r2 missing #2: `python3 bench.py --gpus N` (the driver's plain command) used to exit with 'needs
    torch.distributed.run'.  Now the parent starts the N ranks itself -- before it imports torch or anything that could
    initialise a GPU (a process that has must never be replaced or forked on this pool) -- and returns the worst child's code.
    Here, without a GPU, both children must come up with the rendezvous in their environment, rendezvous over gloo, and refuse
    loudly (no CPU path); the parent must report failure."""
    import ast
    src = open(os.path.join(ROOT, "bench.py")).read()
    tree = ast.parse(src)
    first_torch = min(n.lineno for n in ast.walk(tree) if isinstance(n, (ast.Import, ast.ImportFrom)) and n.col_offset == 0
                      and any(a.name.split(".")[0] in ("torch", "numpy", "so3x") for a in n.names))
    launch = next(n.lineno for n in ast.walk(tree) if isinstance(n, ast.Call) and getattr(n.func, "id", "") == "_self_launch")
    assert launch < first_torch, "the self-launch must come before torch is imported"
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "_self_launch")
    assert not any(isinstance(n, (ast.Import, ast.ImportFrom)) and any(a.name.split(".")[0] == "torch" for a in n.names) for n in ast.walk(fn))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-extras"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0
    assert (r.stdout + r.stderr).count("needs an MI355X") == 2, r.stdout + r.stderr   # both ranks started, met, and refused
    assert "needs torch.distributed.run" not in r.stdout + r.stderr


def test_prepared_cache_sees_writes_that_tensor_versions_miss():
    """so3x.flat.PreparedCache: the cheap key misses `p.data` writes; the buffer's fingerprint finds them when a
    comparison is due (on request, after check_next(), after an idle gap, every N hits) -- and an unchanged buffer never rebuilds."""
    import time
    from so3x.flat import PreparedCache
    flat = torch.arange(1000, dtype=torch.float32)
    built = []
    c = PreparedCache(every=4, idle_s=1e9)
    build = lambda: built.append(1) or len(built)   # noqa: E731
    assert c.get("k", flat, build) == 1 and c.get("k", flat, build) == 1
    flat.data.mul_(2.0)                              # invisible to the key
    assert c.get("k", flat, build) == 1              # no comparison due: stale (hit 2 of 4)
    assert c.get("k", flat, build, check=True) == 2  # on request
    flat.data.add_(1.0)
    c.check_next()
    assert c.get("k", flat, build) == 3              # armed by a mode switch
    flat.data.add_(1.0)
    assert [c.get("k", flat, build) for _ in range(4)][-1] == 4   # every 4th hit
    assert c.get("k2", flat, build) == 5             # a new key rebuilds as before
    c.invalidate()
    assert c.get("k2", flat, build) == 6
    n = len(built)
    for _ in range(20):
        c.get("k2", flat, build, check=True)
    assert len(built) == n                           # unchanged values: comparisons never rebuild
    c2 = PreparedCache(every=10 ** 9, idle_s=0.01)
    c2.get("k", flat, build)
    flat.data.add_(1.0)
    time.sleep(0.03)
    m = len(built)
    c2.get("k", flat, build)
    assert len(built) == m + 1                       # idle gap
