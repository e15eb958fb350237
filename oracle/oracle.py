"""ctypes binding of the CPU oracle (oracle/so3_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Never imported by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libso3_oracle.so")
GOLDEN = os.path.join(_HERE, "..", "tests", "golden")


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("so3_oracle.c", "so3_oracle_impl.h")]
    if force or not os.path.exists(_LIB) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        _lib = C.CDLL(_LIB)
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _arr(x, dt):
    return np.ascontiguousarray(x, dtype=dt)


_DT = {"f32": np.float32, "f64": np.float64}
_CR = {"f32": C.c_float, "f64": C.c_double}

_knots_cache = None


def knots():
    """The reference's fp32 knot / Haar-weight vectors (fixture, distributions.py:15,21)."""
    global _knots_cache
    if _knots_cache is None:
        z = np.load(os.path.join(GOLDEN, "igso3_knots.npz"))
        _knots_cache = (_arr(z["knots"], np.float32), _arr(z["haar_w"], np.float32))
    return _knots_cache


def cosine_beta_schedule(T):
    out = np.empty(T, np.float64)
    lib().so3o_cosine_beta_schedule(C.c_int(T), _p(out))
    return out


SCHED_ROWS = ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
              "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod",
              "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_variance",
              "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2")


def schedule_from_betas(betas):
    betas = _arr(betas, np.float64)
    T = len(betas)
    out = np.empty((12, T), np.float32)
    lib().so3o_schedule_from_betas(_p(betas), C.c_int(T), _p(out))
    return out


def posemb_freqs(half=28):
    out = np.empty(half, np.float32)
    lib().so3o_posemb_freqs(C.c_int(half), _p(out))
    return out


def eps_ft(omega, eps):
    omega = _arr(omega, np.float32).ravel()
    eps = _arr(eps, np.float32).ravel()
    stride = 0 if eps.size == 1 else 1
    out = np.empty_like(omega)
    lib().so3o_eps_ft(_p(omega), _p(eps), C.c_long(stride), _p(out), C.c_long(omega.size))
    return out


def igso3_dlogf(omega, eps):
    omega = _arr(omega, np.float32).ravel()
    eps = _arr(eps, np.float32).ravel()
    stride = 0 if eps.size == 1 else 1
    out = np.empty(omega.size, np.float64)
    lib().so3o_igso3_dlogf(_p(omega), _p(eps), C.c_long(stride), _p(out), C.c_long(omega.size))
    return out


def igso3_build_tables(eps):
    eps = _arr(eps, np.float32).ravel()
    k, w = knots()
    trap = np.empty((eps.size, 999), np.float32)
    lib().so3o_igso3_build_tables(_p(eps), C.c_long(eps.size), _p(k), _p(w), _p(trap))
    return trap


def igso3_log_prob(R, eps):
    R = _arr(R, np.float32).reshape(-1, 9)
    eps = _arr(eps, np.float32).ravel()
    stride = 0 if eps.size == 1 else 1
    out = np.empty(R.shape[0], np.float32)
    lib().so3o_igso3_log_prob(_p(R), _p(eps), C.c_long(stride), _p(out), C.c_long(R.shape[0]))
    return out


def _fn(name, prec):
    return getattr(lib(), f"so3o_{name}_{prec}")


def quat_to_rmat(q, prec="f32"):
    q = _arr(q, _DT[prec]).reshape(-1, 4)
    out = np.empty((q.shape[0], 3, 3), _DT[prec])
    _fn("quat_to_rmat", prec)(_p(q), _p(out), C.c_long(q.shape[0]))
    return out


def log_rmat_vec(R, prec="f32"):
    R = _arr(R, _DT[prec]).reshape(-1, 9)
    out = np.empty((R.shape[0], 3), _DT[prec])
    _fn("log_rmat_vec", prec)(_p(R), _p(out), C.c_long(R.shape[0]))
    return out


def vec2skew(v):
    """util.py:87-92"""
    v = np.asarray(v)
    S = np.zeros(v.shape[:-1] + (3, 3), v.dtype)
    S[..., 2, 1] = v[..., 0]; S[..., 1, 2] = -v[..., 0]
    S[..., 2, 0] = -v[..., 1]; S[..., 0, 2] = v[..., 1]
    S[..., 1, 0] = v[..., 2]; S[..., 0, 1] = -v[..., 2]
    return S


def exp_vec(w, prec="f32"):
    w = _arr(w, _DT[prec]).reshape(-1, 3)
    out = np.empty((w.shape[0], 3, 3), _DT[prec])
    _fn("exp_vec", prec)(_p(w), _p(out), C.c_long(w.shape[0]))
    return out


def so3_scale(R, k, prec="f32"):
    R = _arr(R, _DT[prec]).reshape(-1, 9)
    k = _arr(k, _DT[prec]).ravel()
    stride = 0 if k.size == 1 else 1
    out = np.empty((R.shape[0], 3, 3), _DT[prec])
    _fn("so3_scale", prec)(_p(R), _p(k), C.c_long(stride), _p(out), C.c_long(R.shape[0]))
    return out


def aa_to_rmat(axis, ang, prec="f32"):
    axis = _arr(axis, _DT[prec]).reshape(-1, 3)
    ang = _arr(ang, _DT[prec]).ravel()
    out = np.empty((axis.shape[0], 3, 3), _DT[prec])
    _fn("aa_to_rmat", prec)(_p(axis), _p(ang), _p(out), C.c_long(axis.shape[0]))
    return out


def rmat_to_aa(R, prec="f32"):
    R = _arr(R, _DT[prec]).reshape(-1, 9)
    n = R.shape[0]
    axis = np.empty((n, 3), _DT[prec])
    ang = np.empty((n, 1), _DT[prec])
    _fn("rmat_to_aa", prec)(_p(R), _p(axis), _p(ang), C.c_long(n))
    return axis, ang


def so3_lerp(a, b, w, prec="f32"):
    b = _arr(b, _DT[prec]).reshape(-1, 9)
    a = _arr(a, _DT[prec]).reshape(-1, 9)
    w = _arr(w, _DT[prec]).ravel()
    n = b.shape[0]
    out = np.empty((n, 3, 3), _DT[prec])
    _fn("so3_lerp", prec)(_p(a), C.c_long(0 if a.shape[0] == 1 else 9), _p(b), _p(w),
                          C.c_long(0 if w.size == 1 else 1), _p(out), C.c_long(n))
    return out


def rmat_dist(a, b, prec="f32"):
    a = _arr(a, _DT[prec]).reshape(-1, 9)
    b = _arr(b, _DT[prec]).reshape(-1, 9)
    out = np.empty(a.shape[0], _DT[prec])
    _fn("rmat_dist", prec)(_p(a), _p(b), _p(out), C.c_long(a.shape[0]))
    return out


def igso3_sample(trap, axes, unif, row_idx=None, weight_row=-1, mean=None, prec="f32"):
    """trap [n_rows,999]; returns (rotations [n,3,3], angles [n])."""
    trap = _arr(trap, np.float32).reshape(-1, 999)
    axes = _arr(axes, np.float32).reshape(-1, 3)
    unif = _arr(unif, np.float32).ravel()
    n = unif.size
    k, _ = knots()
    ri = _arr(row_idx, np.int64) if row_idx is not None else None
    m = _arr(mean, _DT[prec]) if mean is not None else None
    out = np.empty((n, 3, 3), _DT[prec])
    ang = np.empty(n, np.float32)
    _fn("igso3_sample", prec)(_p(trap), _p(ri), C.c_long(weight_row), _p(k), _p(axes), _p(unif), _p(m),
                              _p(out), _p(ang), C.c_long(n))
    return out, ang


N_PARAMS = 4 * (65 * 65 + 65) + 3 * 65 + 3
N_PARAMS_ROTMAT = 4 * (65 * 65 + 65) + 6 * 65 + 6   # out_type="rotmat": a 6-wide head (so3_train.py:19-22)


def _head_width(n_params, trunk):
    """3 ("skewvec") or 6 ("rotmat") outputs, read off the flat parameter count"""
    for k in (3, 6):
        if n_params == trunk + k * (1 + (65 if trunk == 4 * (65 * 65 + 65) else 255)):
            return k
    raise ValueError(f"unexpected parameter count {n_params}")


def mlp_fwd(params, R, t, prec="f32", return_acts=False):
    params = _arr(params, np.float32).ravel()
    nout = _head_width(params.size, 4 * (65 * 65 + 65))
    R = _arr(R, _DT[prec]).reshape(-1, 9)
    t = _arr(t, np.int64).ravel()
    n = R.shape[0]
    out = np.empty((n, nout), _DT[prec])
    acts = np.empty((n, 5, 65), _DT[prec]) if return_acts else None
    fr = posemb_freqs()
    _fn("mlp_fwd", prec)(_p(params), _p(fr), _p(R), _p(t), C.c_long(0 if t.size == 1 else 1), _p(out),
                         _p(acts), C.c_long(n), C.c_long(nout))
    return (out, acts) if return_acts else out


def mlp_bwd(params, R, t, dout, prec="f32"):
    params = _arr(params, np.float32).ravel()
    R = _arr(R, _DT[prec]).reshape(-1, 9)
    t = _arr(t, np.int64).ravel()
    nout = _head_width(params.size, 4 * (65 * 65 + 65))
    dout = _arr(dout, _DT[prec]).reshape(-1, nout)
    n = R.shape[0]
    dp = np.empty(params.size, np.float64)
    fr = posemb_freqs()
    _fn("mlp_bwd", prec)(_p(params), _p(fr), _p(R), _p(t), C.c_long(0 if t.size == 1 else 1), _p(dout),
                         _p(dp), C.c_long(n), C.c_long(nout))
    return dp


N_PARAMS_RESNET = 6 * (255 * 255 + 255) + 3 * 255 + 3


def resnet_fwd(params, R, t, prec="f32"):
    """so3_lock_train.RotPredict(out_type="skewvec") forward (so3_lock_train.py:11-59)."""
    params = _arr(params, np.float32).ravel()
    nout = _head_width(params.size, 6 * (255 * 255 + 255))
    R = _arr(R, _DT[prec]).reshape(-1, 9)
    t = _arr(t, np.int64).ravel()
    n = R.shape[0]
    out = np.empty((n, nout), _DT[prec])
    fr = posemb_freqs(123)
    _fn("resnet_fwd", prec)(_p(params), _p(fr), _p(R), _p(t), C.c_long(0 if t.size == 1 else 1), _p(out), C.c_long(n),
                            C.c_long(nout))
    return out


def resnet_bwd(params, R, t, dout, prec="f32"):
    params = _arr(params, np.float32).ravel()
    R = _arr(R, _DT[prec]).reshape(-1, 9)
    t = _arr(t, np.int64).ravel()
    nout = _head_width(params.size, 6 * (255 * 255 + 255))
    dout = _arr(dout, _DT[prec]).reshape(-1, nout)
    dp = np.empty(params.size, np.float64)
    fr = posemb_freqs(123)
    _fn("resnet_bwd", prec)(_p(params), _p(fr), _p(R), _p(t), C.c_long(0 if t.size == 1 else 1), _p(dout), _p(dp),
                            C.c_long(R.shape[0]), C.c_long(nout))
    return dp


N_PARAMS_RESNET_ROTMAT = 6 * (255 * 255 + 255) + 6 * 255 + 6


def six2rmat(x, prec="f32"):
    """util.py:67-76"""
    x = _arr(x, _DT[prec]).reshape(-1, 6)
    out = np.empty((x.shape[0], 3, 3), _DT[prec])
    _fn("six2rmat", prec)(_p(x), _p(out), C.c_long(x.shape[0]))
    return out


def six2rmat_bwd(x, G, prec="f32"):
    """autograd of six2rmat: dL/dout [n,3,3] -> dL/dx [n,6]"""
    x = _arr(x, _DT[prec]).reshape(-1, 6)
    G = _arr(G, _DT[prec]).reshape(-1, 9)
    dx = np.empty_like(x)
    _fn("six2rmat_bwd", prec)(_p(x), _p(G), _p(dx), C.c_long(x.shape[0]))
    return dx


def log_rmat_bwd(R, G, prec="f32"):
    """autograd of log_rmat (util.py:164-175): dL/dlog [n,3,3] -> dL/dR"""
    R = _arr(R, _DT[prec]).reshape(-1, 9)
    G = _arr(G, _DT[prec]).reshape(-1, 9)
    dR = np.empty_like(R)
    _fn("log_rmat_bwd", prec)(_p(R), _p(G), _p(dR), C.c_long(R.shape[0]))
    return dR.reshape(-1, 3, 3)


def rmat_dist_bwd(a, b, g, prec="f32"):
    """autograd of rmat_dist (util.py:315-322): dL/ddist [n] -> (dL/da, dL/db)"""
    a = _arr(a, _DT[prec]).reshape(-1, 9)
    b = _arr(b, _DT[prec]).reshape(-1, 9)
    g = _arr(g, _DT[prec]).ravel()
    da, db = np.empty_like(a), np.empty_like(b)
    _fn("rmat_dist_bwd", prec)(_p(a), _p(b), _p(g), _p(da), _p(db), C.c_long(a.shape[0]))
    return da.reshape(-1, 3, 3), db.reshape(-1, 3, 3)


def prevstep_loss(x_recon, x_start, x_noisy, sched, t, prec="f32"):
    """loss_type="prevstep" (diffusion.py:358-365): returns (step [n,3,3], dist2 [n], d(sum dist2)/dx_recon [n,3,3]);
    the loss is dist2.mean().  sched = schedule_from_betas table (rows 10/11 = posterior_mean_coef1/2)."""
    xr = _arr(x_recon, _DT[prec]).reshape(-1, 9)
    xs = _arr(x_start, _DT[prec]).reshape(-1, 9)
    xn = _arr(x_noisy, _DT[prec]).reshape(-1, 9)
    t = _arr(t, np.int64).ravel()
    n = xr.shape[0]
    c1, c2 = np.ascontiguousarray(sched[10], np.float32), np.ascontiguousarray(sched[11], np.float32)
    step, d2, dx = np.empty_like(xr), np.empty(n, _DT[prec]), np.empty_like(xr)
    _fn("prevstep_loss", prec)(_p(xr), _p(xs), _p(xn), _p(c1), _p(c2), _p(t), _p(step), _p(d2), _p(dx), C.c_long(n))
    return step.reshape(-1, 3, 3), d2, dx.reshape(-1, 3, 3)


def q_sample_target(x0, noise, sched, t, prec="f32"):
    """returns (x_t, target); sched = schedule_from_betas table."""
    x0 = _arr(x0, _DT[prec]).reshape(-1, 9)
    noise = _arr(noise, _DT[prec]).reshape(-1, 9)
    t = _arr(t, np.int64).ravel()
    n = x0.shape[0]
    xt = np.empty((n, 3, 3), _DT[prec])
    tg = np.empty((n, 3), _DT[prec])
    s3 = _arr(sched[3], np.float32)
    s4 = _arr(sched[4], np.float32)
    _fn("q_sample_target", prec)(_p(x0), _p(noise), _p(s3), _p(s4), _p(t), _p(xt), _p(tg), C.c_long(n))
    return xt, tg


def p_mean(x, v, a, b, c1, c2, prec="f32"):
    """returns (x0hat, mean) for one shared timestep's coefficients."""
    x = _arr(x, _DT[prec]).reshape(-1, 9)
    v = _arr(v, _DT[prec]).reshape(-1, 3)
    n = x.shape[0]
    x0h = np.empty((n, 3, 3), _DT[prec])
    mean = np.empty((n, 3, 3), _DT[prec])
    cr = _CR[prec]
    f = _fn("p_mean", prec)
    f.argtypes = [C.c_void_p, C.c_void_p, cr, cr, cr, cr, C.c_void_p, C.c_void_p, C.c_long]
    f(_p(x), _p(v), cr(a), cr(b), cr(c1), cr(c2), _p(x0h), _p(mean), n)
    return x0h, mean


def rmul(a, b, prec="f32"):
    a = _arr(a, _DT[prec]).reshape(-1, 9)
    b = _arr(b, _DT[prec]).reshape(-1, 9)
    out = np.empty((a.shape[0], 3, 3), _DT[prec])
    _fn("rmul", prec)(_p(a), _p(b), _p(out), C.c_long(a.shape[0]))
    return out


def domega_dR(R, prec="f32"):
    R = _arr(R, _DT[prec]).reshape(-1, 9)
    out = np.empty((R.shape[0], 3, 3), _DT[prec])
    _fn("domega_dR", prec)(_p(R), _p(out), C.c_long(R.shape[0]))
    return out


def p_sample_step(params, sched, trap_p, x, t, axes, unif):
    """fp32, OpenMP over the batch: the timed CPU baseline and chain checker."""
    params = _arr(params, np.float32).ravel()
    sched = _arr(sched, np.float32)
    T = sched.shape[1]
    trap_p = _arr(trap_p, np.float32).reshape(T, 999)
    x = _arr(x, np.float32).reshape(-1, 9)
    n = x.shape[0]
    axes = _arr(axes, np.float32) if axes is not None else np.zeros((n, 3), np.float32)
    unif = _arr(unif, np.float32) if unif is not None else np.zeros(n, np.float32)
    out = np.empty((n, 3, 3), np.float32)
    k, _ = knots()
    fr = posemb_freqs()
    lib().so3o_p_sample_step_f32(_p(params), _p(fr), _p(sched), C.c_int(T), _p(trap_p), _p(k), _p(x),
                                 C.c_int(int(t)), _p(axes), _p(unif), _p(out), C.c_long(n))
    return out


def omp_threads():
    return int(lib().so3o_omp_threads())


FAST_FLAGS = "-O3 -march=native -fopenmp (contraction allowed)"
_fast = None


def p_sample_step_timed(params, sched, trap_p, x, t, axes, unif, out):
    """so3o_p_sample_step_f32 of the TIMED build (the same source compiled FAST_FLAGS ON THIS HOST -- the GPU box's cores differ
    from the build container's, so it is built where it runs) on preallocated, contiguous fp32 arrays: nothing but the C call
    inside the caller's timed region.  bench.py's cpu_baseline leg only; never a checker.  The build goes to a per-process file
    under the system's temporary directory (a read-only tree or two concurrent runs cannot break it); if no compiler is there,
    the prebuilt -O2 checker library is timed instead and FAST_FLAGS says so."""
    global _fast, FAST_FLAGS
    if _fast is None:
        import tempfile
        path = os.path.join(tempfile.gettempdir(), f"libso3_oracle_fast.{os.getpid()}.so")
        try:
            subprocess.check_call([os.environ.get("CC", "gcc"), "-O3", "-march=native", "-fPIC", "-fopenmp", "-fno-fast-math", "-Wall",
                                   "-Wno-unused-function", "-shared", "-o", path, os.path.join(_HERE, "so3_oracle.c"), "-lm"])
            _fast = C.CDLL(path)
        except (OSError, subprocess.CalledProcessError) as e:
            FAST_FLAGS = f"-O2 -ffp-contract=off (the prebuilt checker library: the -O3 -march=native build failed here: {e!r})"
            _fast = lib()
        finally:
            try:
                os.unlink(path)      # the mapping stays valid
            except OSError:
                pass
    T = sched.shape[1]
    k, _ = knots()
    fr = posemb_freqs()
    _fast.so3o_p_sample_step_f32(_p(params), _p(fr), _p(sched), C.c_int(T), _p(trap_p), _p(k), _p(x), C.c_int(int(t)), _p(axes), _p(unif),
                                 _p(out), C.c_long(x.shape[0]))


def flat_params(state):
    """state: mapping with net_{0,2,4,6,8}_{weight,bias} (tests/golden/score_mlp.npz naming)."""
    parts = []
    for l in (0, 2, 4, 6, 8):
        parts.append(np.asarray(state[f"net_{l}_weight"], np.float32).ravel())
        parts.append(np.asarray(state[f"net_{l}_bias"], np.float32).ravel())
    return np.concatenate(parts)


# ---------------------------------------------------------------------------------------------
# Sample-quality statistics used by gate G3 (numpy restatement; reference util.py:128-134, 254-299)
# ---------------------------------------------------------------------------------------------
def pairwise_rmat_dist(X, Y):
    """rmat_dist(x_i, y_j) for all pairs: ||log(x_i^T y_j)||_F = sqrt(2) * angle  (util.py:315-322)."""
    X = np.asarray(X, np.float64).reshape(-1, 3, 3)
    Y = np.asarray(Y, np.float64).reshape(-1, 3, 3)
    # M = x^T y ; M[a][b] = sum_k x[k][a] y[k][b]
    def M(a, b):
        return X[:, :, a] @ Y[:, :, b].T
    tr = M(0, 0) + M(1, 1) + M(2, 2)
    v0 = M(2, 1) - M(1, 2)
    v1 = M(0, 2) - M(2, 0)
    v2 = M(1, 0) - M(0, 1)
    s = np.sqrt(v0 * v0 + v1 * v1 + v2 * v2) / 2
    c = (tr - 1) / 2
    return np.sqrt(2.0) * np.arctan2(s, c)


def rmat_gaussian_kernel_matrix(X, Y):
    """exp(-rmat_dist) (util.py:128-134)"""
    return np.exp(-pairwise_rmat_dist(X, Y))


def MMD(X, Y):
    """Maximum mean discrepancy with the reference's estimator (util.py:254-286): means over ALL pairs."""
    lx, ly = len(X), len(Y)
    return (rmat_gaussian_kernel_matrix(X, X).sum() / lx ** 2 + rmat_gaussian_kernel_matrix(Y, Y).sum() / ly ** 2
            - 2 * rmat_gaussian_kernel_matrix(X, Y).sum() / (lx * ly))


def ker_2samp_threshold(m, alpha=0.05, max_ker=1.0):
    """acceptance bound of Ker_2samp_test (util.py:289-299)"""
    return (2 * max_ker / m) ** 0.5 * (1 + (2 * np.log(1 / alpha)) ** 0.5)


def Ker_2samp_test(X, Y, alpha=0.05):
    assert len(X) == len(Y)
    return MMD(X, Y) < ker_2samp_threshold(len(X), alpha)


# ---------------------------------------------------------------------------------------------
# SE(3) layer: numpy restatement on top of the C rotation oracle
# ---------------------------------------------------------------------------------------------
def se3_q_sample_target(x0_rot, x0_shift, noise_rot, noise_shift, sched, t, shift_scale, prec="f32"):
    """SE3Diffusion.q_sample (diffusion.py:496-503) and the p_losses targets (:511-512)."""
    dt = _DT[prec]
    t = np.asarray(t, np.int64)
    xt_rot, tg_rot = q_sample_target(x0_rot, noise_rot, sched, t, prec)
    k = sched[3][t].astype(dt)[:, None]
    eps = sched[4][t].astype(dt)[:, None]
    ns = np.asarray(noise_shift, dt)
    xt_shift = np.asarray(x0_shift, dt) * k + ns                     # se3_scale shift (util.py:384) + noise.shift
    tg_shift = ns * (1 / (eps * dt(shift_scale)))
    return xt_rot, xt_shift, tg_rot, tg_shift


def se3_p_mean(x_rot, x_shift, v_rot, v_shift, sched, t, prec="f32"):
    """SE3Diffusion.predict_start_from_noise + q_posterior (diffusion.py:444-466), one shared t."""
    dt = _DT[prec]
    a, b, c1, c2 = (dt(sched[i][t]) for i in (6, 7, 10, 11))
    _, mean_rot = p_mean(x_rot, v_rot, float(a), float(b), float(c1), float(c2), prec)
    xs = np.asarray(x_shift, dt)
    x0h = xs * a - np.asarray(v_shift, dt) * b
    return mean_rot, x0h * c1 + xs * c2


def move_prot(rot, shift, pos, frames):
    """prot_util.py:73-81, batched over structures: pos [S,L,3], frames [S,L,3,3]."""
    rot = np.asarray(rot, np.float64); shift = np.asarray(shift, np.float64)
    pos = np.asarray(pos, np.float64); frames = np.asarray(frames, np.float64)
    mean = pos.mean(axis=-2, keepdims=True)
    RT = np.swapaxes(rot, -1, -2)[:, None]
    out_pos = ((pos - mean)[:, :, None, :] @ RT)[:, :, 0, :] + mean + shift[:, None, :]
    out_fr = frames @ RT
    return out_pos, out_fr


# ---------------------------------------------------------------------------------------------
# the optimizer update of the training loop (so3_train.py:64,76)
# ---------------------------------------------------------------------------------------------
def adam_step(p, g, m, v, step, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, grad_scale=1.0):
    """torch.optim.Adam.step() on flat fp32 arrays; returns the new (p, m, v).  `step` = the count before the call."""
    p, m, v = (_arr(a, np.float32).copy() for a in (p, m, v))
    g = _arr(g, np.float32)
    lib().so3o_adam_step(_p(p), _p(g), _p(m), _p(v), C.c_long(p.size), C.c_double(step), C.c_double(lr), C.c_double(betas[0]),
                         C.c_double(betas[1]), C.c_double(eps), C.c_double(weight_decay), C.c_double(grad_scale))
    return p, m, v


def orthogonalise(mat):
    """util.py:95-107: U round(S) V^T of the SVD of the leading 3x3 block (numpy float64 SVD; torch.round = half to even)"""
    mat = np.asarray(mat)
    out = mat.astype(np.float64).copy()
    u, s, vt = np.linalg.svd(out[..., :3, :3])
    out[..., :3, :3] = (u * np.rint(s)[..., None, :]) @ vt
    return out.astype(mat.dtype)
