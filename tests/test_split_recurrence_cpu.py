"""CPU: the arithmetic a list-splitting blend-backward would rest on (DESIGN.md section 8 "Next" #1; not built yet).

The backward walks a pixel's composited list back to front with two scalars of state (csrc/surfel_backward.hip: pair_gradients):
    T_k = T_final / prod_{i >= k} (1 - alpha_i)                      transmittance in front of instance k
    X_k = T_final bg.gC + sum_{i > k} w_i u_i ,  w_i = alpha_i T_i ,  u_i = c_i.gC + d_i g_D + g_A + n_i.gN + g_dist (mm_i^2 A - 2 mm_i M1 + M2)
    dL/dalpha_k = T_k u_k - X_k / (1 - alpha_k)
A workgroup that starts in the MIDDLE of the list (position s) needs (T_s+1, X_s).  The forward can leave, per pixel and boundary, its
prefix state (T, C[3], D, N[3], M1, M2 — ten floats); then
    X_s = T_final bg.gC + gC.(C_f - C_s) + g_D (D_f - D_s) + gN.(N_f - N_s) + g_A (T_s - T_f)
          + g_dist (A (M2_f - M2_s) - 2 M1 (M1_f - M1_s) + M2 (T_s - T_f))
with every quantity a difference of FORWARD sums.  This test replays both formulations in float32 on random lists and shows that the
differences lose nothing that matters: dL/dalpha of the split walk agrees with the sequential walk to the fp32 tolerance the parity
tests use (1e-4 of the mean magnitude + 2e-3 relative), also when the tail behind the boundary carries a tiny share of the sum."""
import numpy as np
import pytest

F = np.float32


class _Prec:
    """the arithmetic type of forward() / walk(): float32 (the kernels') or float64 (the yardstick)"""
    t = np.float32


def forward(alpha, c, d, nrm, mm):
    """sequential front-to-back compositing of one pixel; returns final sums and the prefix state after every position"""
    F = _Prec.t
    T = F(1.0)
    C = np.zeros(3, F); N = np.zeros(3, F); D = F(0); M1 = F(0); M2 = F(0)
    pre = []
    for a, ci, di, ni, mi in zip(alpha, c, d, nrm, mm):
        w = F(a * T)
        C = (C + ci * w).astype(F); N = (N + ni * w).astype(F); D = F(D + di * w); M1 = F(M1 + mi * w); M2 = F(M2 + mi * mi * w)
        T = F(T * (F(1) - a))
        pre.append((T, C.copy(), D, N.copy(), M1, M2))
    return pre


def walk(alpha, c, d, nrm, mm, g, fin, lo, hi, T_top, X_top):
    """back-to-front walk over positions [lo, hi) starting from (T after hi - 1, X behind hi - 1); dL/dalpha per position"""
    F = _Prec.t
    gC, gD, gA, gN, gdist, bg = g
    Tf, Cf, Df, Nf, M1f, M2f = fin
    A = F(1) - Tf
    T, X = F(T_top), F(X_top)
    out = np.zeros(hi - lo, F)
    for k in range(hi - 1, lo - 1, -1):
        i1a = F(1) / (F(1) - alpha[k])
        T = F(T * i1a)
        w = F(alpha[k] * T)
        u = F(gdist * (mm[k] * (mm[k] * A - F(2) * M1f) + M2f) + gA)
        u = F(u + c[k] @ gC + d[k] * gD + nrm[k] @ gN)
        out[k - lo] = F(T * u - X * i1a)
        X = F(X + w * u)
    return out


@pytest.mark.parametrize("seed,n,split,alpha_hi", [(0, 400, 256, 0.05), (1, 700, 256, 0.02), (2, 300, 128, 0.3), (3, 600, 512, 0.01)])
def test_split_walk_matches_the_sequential_walk(seed, n, split, alpha_hi):
    rng = np.random.default_rng(seed)
    alpha = rng.uniform(1.0 / 255.0, alpha_hi, n).astype(F)
    c = rng.uniform(0, 1, (n, 3)).astype(F); nrm = rng.normal(size=(n, 3)).astype(F)
    d = rng.uniform(0.5, 9.0, n).astype(F)
    mm = (F(100.0 / 99.8) - F(100.0 / 99.8 * 0.2) / d).astype(F)
    g = (rng.normal(size=3).astype(F), F(rng.normal()), F(rng.normal()), rng.normal(size=3).astype(F), F(abs(rng.normal()) * 100), rng.uniform(0, 1, 3).astype(F))
    gC, gD, gA, gN, gdist, bg = g

    def both(prec):
        _Prec.t = prec
        P = prec
        cast = lambda x: np.asarray(x).astype(P) if isinstance(x, np.ndarray) else P(x)
        al, cc, dd, nn, mmm = cast(alpha), cast(c), cast(d), cast(nrm), cast(mm)
        gg = tuple(cast(x) for x in g)
        gC_, gD_, gA_, gN_, gdist_, bg_ = gg
        pre = forward(al, cc, dd, nn, mmm)
        fin = pre[-1]
        Tf, Cf, Df, Nf, M1f, M2f = fin
        X_end = P(Tf * (bg_ @ gC_))
        seq = walk(al, cc, dd, nn, mmm, gg, fin, 0, n, Tf, X_end)                    # one workgroup, the whole list
        hi_part = walk(al, cc, dd, nn, mmm, gg, fin, split, n, Tf, X_end)             # second segment: starts like the whole walk
        Ts, Cs, Ds, Ns, M1s, M2s = pre[split - 1]                                     # forward checkpoint behind position `split`
        A = P(1) - Tf
        X_s = P(X_end + (Cf - Cs) @ gC_ + (Df - Ds) * gD_ + (Nf - Ns) @ gN_ + gA_ * (Ts - Tf)
                + gdist_ * (A * (M2f - M2s) - P(2) * M1f * (M1f - M1s) + M2f * (Ts - Tf)))
        lo_part = walk(al, cc, dd, nn, mmm, gg, fin, 0, split, Ts, X_s)               # first segment: from the checkpoint
        assert np.array_equal(hi_part, seq[split:])
        return seq, np.concatenate([lo_part, hi_part]), Ts, Tf
    try:
        ref, split64, Ts64, _ = both(np.float64)
        seq, got, Ts, Tf = both(np.float32)
    finally:
        _Prec.t = np.float32
    assert np.allclose(split64, ref, rtol=1e-6, atol=1e-9 * np.abs(ref).mean())       # the identity itself (fp64: the differences cancel to rounding)
    tol = 1e-4 * np.abs(ref).mean() + 2e-3 * np.abs(ref)
    f_seq, f_split = (np.abs(seq - ref) <= tol).mean(), (np.abs(got - ref) <= tol).mean()
    # the split walk in fp32 is held to the parity tolerance against the fp64 walk, and may not fall behind the sequential fp32 walk
    assert f_split >= 0.97 and f_split >= f_seq - 0.03, (f_seq, f_split, np.abs(got - ref).max(), np.abs(seq - ref).max())
    # the transmittance the first segment starts from is the forward's own product: no worse than dividing T_final back up the list
    T_div = np.float32(Tf)
    for k in range(n - 1, split - 1, -1):
        T_div = np.float32(T_div / (np.float32(1) - alpha[k]))
    assert abs(float(Ts) - float(Ts64)) <= abs(float(T_div) - float(Ts64)) + 1e-6 * float(Ts64)
