"""ORACLE — ctypes wrapper for exact_chain.c (test infrastructure only)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libaew_oracle.so")
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "exact_chain.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE] + (["-B"] if force else []) + ["all"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(ctypes.c_void_p)


KSPLIT = 1          # the canonical order of the exact chains: 1 = one ascending chain per output (the product's default);
                    # 2 | 4 = the split form of aew_gemm_nt_t.k_split - must equal EncoderPlan.k_split (0 / 1 = off) of the engine
                    # it is compared with (tests/test_gpu_parity.py::test_encoder_split_k_is_exact_in_its_own_order sets both)


def ksplit_for(k_total_padded: int, S: int = None) -> int:
    """S if the op's K axis (taps x input channels PADDED to the kernel's 64-channel granule) is a multiple of 32 * S and the
    input channels need no padding, else 1 - the rule the engine applies to every exact fp32 GEMM (encoder layers,
    bottleneck linear)."""
    S = KSPLIT if S is None else S
    return S if (S > 1 and k_total_padded % (32 * S) == 0) else 1


def conv_cl(x, W, bias, stride=1, relu=True, res_lw=-1, ksplit=1):
    """x (B, L, Cin) channels-last; W (Cout, Cin, f) reference layout; -> (B, Lout, Cout).  ksplit: see exact_chain.c."""
    x, xp = _f(x)
    W, wp = _f(W)
    B, L, Cin = x.shape
    Cout, Cin2, f = W.shape
    assert Cin == Cin2
    Lout = (L - f) // stride + 1
    y = np.empty((B, Lout, Cout), np.float32)
    bp = None
    if bias is not None:
        bias, bp = _f(bias)
    rc = lib().aewo_conv_cl_split(xp, B, L, Cin, wp, bp, Cout, f, stride, int(relu), int(res_lw),
                                  y.ctypes.data_as(ctypes.c_void_p), int(ksplit))
    assert rc == 0, rc
    return y


ENC_FILTERS = (3, 3, 4, 3, 3, 1, 1, 1, 1)
ENC_STRIDES = (1, 1, 2, 1, 1, 1, 1, 1, 1)
ENC_RESIDUAL = (False, True, False, True, True, True, True, True, True)


def encoder_cl(sd, pre, mel_cl):
    """Encoder stack on channels-last mel (B, F, n_in) -> (B, N_e, n_out)."""
    x = np.ascontiguousarray(mel_cl, np.float32)
    for i, (f, s, r) in enumerate(zip(ENC_FILTERS, ENC_STRIDES, ENC_RESIDUAL)):
        W = np.asarray(sd[f"{pre}net.{i}.conv.weight"])
        cin = W.shape[1]
        # (input channels that the kernel pads - the 39 mel channels of layer 0, reduced-width test models - are not split:
        # the padded k axis would cut elsewhere than the unpadded one)
        ks = ksplit_for(f * cin) if cin % 64 == 0 else 1
        x = conv_cl(x, W, np.asarray(sd[f"{pre}net.{i}.conv.bias"]), s, True, (f - 1) // 2 if r else -1, ksplit=ks)
    return x


def linear_cl(x_cl, W):
    """1x1 conv without bias: (B, N, Cin) x (Cout, Cin, 1)."""
    W = np.asarray(W)
    ks = ksplit_for(W.shape[1] * W.shape[2]) if W.shape[1] % 64 == 0 else 1
    return conv_cl(x_cl, W, None, 1, False, -1, ksplit=ks)


def vq_nearest(ze_q, emb, metric="scaled_l2"):
    """ze_q (Q, d), emb (K, d) -> ind int64 (Q,), dist (Q,), second-best dist (Q,)."""
    ze_q, zp = _f(ze_q)
    emb, ep = _f(emb)
    Q, d = ze_q.shape
    K = emb.shape[0]
    ind = np.empty(Q, np.int64)
    dist = np.empty(Q, np.float32)
    sec = np.empty(Q, np.float32)
    rc = lib().aewo_vq_nearest(zp, ep, Q, K, d, 0 if metric == "scaled_l2" else 1,
                               ind.ctypes.data_as(ctypes.c_void_p),
                               dist.ctypes.data_as(ctypes.c_void_p),
                               sec.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0
    return ind, dist, sec


def vq_stats(ze_q, ind, K):
    ze_q, zp = _f(ze_q)
    ind = np.ascontiguousarray(ind, np.int64)
    Q, d = ze_q.shape
    z_sum = np.empty((K, d), np.float32)
    n_sum = np.empty(K, np.float32)
    rc = lib().aewo_vq_stats(zp, ind.ctypes.data_as(ctypes.c_void_p), Q, K, d,
                             z_sum.ctypes.data_as(ctypes.c_void_p),
                             n_sum.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0
    return z_sum, n_sum


def ema(numer, denom, z_sum, n_sum, gamma):
    numer = np.array(numer, np.float32, copy=True)
    denom = np.array(denom, np.float32, copy=True)
    z_sum, zp = _f(z_sum)
    n_sum, np_ = _f(n_sum)
    K, d = numer.shape
    rc = lib().aewo_ema(numer.ctypes.data_as(ctypes.c_void_p), denom.ctypes.data_as(ctypes.c_void_p),
                        zp, np_, K, d, ctypes.c_double(gamma))
    assert rc == 0
    return numer, denom


def codebook(numer, denom):
    numer, a = _f(numer)
    denom, b = _f(denom)
    emb = np.empty_like(numer)
    rc = lib().aewo_codebook(a, b, emb.ctypes.data_as(ctypes.c_void_p), numer.shape[0], numer.shape[1])
    assert rc == 0
    return emb
