"""CPU-only accuracy probe: Winograd F(2x2, 3x3) against the direct form, both with the engine's fp16 x 2 operand split, three MFMA
products and fp32 accumulation, at the ResnetBlock shape (512 input channels, 3 x 3, 32 x 32 pixels; 64 of the 512 output channels).

MFMA model (as tests/emu): the 16 products of a k-group are exact (fp16 x fp16 fits fp32), their sum and the accumulator are added in
high precision and rounded to fp32 ONCE per instruction.  Direct: K order slab-major, chains of two 16-channel slabs (54 instructions)
folded into a running fp32 total -- conv_h2.hpp.  Winograd: input transform V = B^T d B on the fp32 activations (after the producer's
InstanceNorm + ReLU), THEN the split (the transform of a split plane is not a split: sums of four fp16 numbers need more bits); filter
transform U = G g G^T in fp64 at pack time, then scale + split; 16 element-wise GEMMs over the 512 channels (96 instructions each: one
chain, or chains of 8 k-groups folded into a total); output transform Y = A^T M A in fp32.

Reports mean / max |error| against the fp64 convolution, relative to max |ref|.  The go / no-go line of VERDICT r3 #2: Winograd <= 1.5 x
the direct error.
    python tools/probes/winograd_probe.py [seeds]"""
import sys

import numpy as np

f32, f64 = np.float32, np.float64


def split16(x):
    """fp32 array (already scaled) -> (hi, lo) fp16-valued fp32 arrays"""
    hi = x.astype(np.float16)
    lo = (x - hi.astype(f32)).astype(np.float16)
    return hi.astype(f64), lo.astype(f64)


def scale_for(amax):
    e = int(np.floor(np.log2(amax))) + 1
    return 2.0 ** (15 - e)


def mfma_chain(A_hi, A_lo, B_hi, B_lo, groups, chain_groups):
    """sum over k of (A_hi+A_lo)[m,k] * (B_hi+B_lo)[k,n] the engine's way: per 16-wide k-group three instructions lo*hi, hi*lo, hi*hi;
    `groups` = list of k-group index arrays in K order; chains of `chain_groups` groups are folded into a running total (0 = one chain)"""
    M, N = A_hi.shape[0], B_hi.shape[1]
    tot = np.zeros((M, N), f32)
    acc = np.zeros((M, N), f32)
    for gi, g in enumerate(groups):
        for a, b in ((A_lo, B_hi), (A_hi, B_lo), (A_hi, B_hi)):
            acc = (acc.astype(f64) + a[:, g] @ b[g, :]).astype(f32)
        if chain_groups and (gi + 1) % chain_groups == 0:
            tot = (tot + acc).astype(f32); acc = np.zeros((M, N), f32)
    if chain_groups == 0:
        return acc
    if len(groups) % chain_groups:
        tot = (tot + acc).astype(f32)
    return tot


def run(seed, C=512, Co=64, H=32, W=32):
    rng = np.random.default_rng(seed)
    x = np.maximum(rng.standard_normal((C, H, W)), 0).astype(f32)              # relu(InstanceNorm(.)): the consumer side of a ResnetBlock
    w = (rng.uniform(-1, 1, (Co, C, 3, 3)) * 2.0 / np.sqrt(C * 9)).astype(f32)
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1)), mode="reflect")
    # fp64 reference
    ref = np.zeros((Co, H, W), f64)
    for ky in range(3):
        for kx in range(3):
            ref += np.einsum("oc,chw->ohw", w[:, :, ky, kx].astype(f64), xp[:, ky:ky + H, kx:kx + W].astype(f64))
    rmax = np.abs(ref).max()
    out = {}

    # ---- direct, the engine's arithmetic
    sa, sw = scale_for(np.abs(x).max()), scale_for(np.abs(w).max())
    xh, xl = split16((xp * f32(sa)).astype(f32))
    wh, wl = split16((w * f32(sw)).astype(f32))
    A_hi = np.concatenate([xh[:, ky:ky + H, kx:kx + W].reshape(C, -1).T for kx in range(3) for ky in range(3)], axis=1)      # [P, 9C], tap-major blocks
    A_lo = np.concatenate([xl[:, ky:ky + H, kx:kx + W].reshape(C, -1).T for kx in range(3) for ky in range(3)], axis=1)
    B_hi = np.concatenate([wh[:, :, ky, kx].T for kx in range(3) for ky in range(3)], axis=0)                               # [9C, Co]
    B_lo = np.concatenate([wl[:, :, ky, kx].T for kx in range(3) for ky in range(3)], axis=0)
    groups = [np.arange(t * C + s * 16, t * C + s * 16 + 16) for s in range(C // 16) for t in range(9)]                     # slab-major, 9 taps per slab
    for nm, cg in (("direct, chains of 2 slabs (product)", 18), ("direct, one chain", 0)):
        y = mfma_chain(A_hi, A_lo, B_hi, B_lo, groups, cg).astype(f64) / (sa * sw)
        e = np.abs(y.T.reshape(Co, H, W) - ref)
        out[nm] = (e.mean() / rmax, e.max() / rmax)
    # exact-fp32 chain for scale: fl32 accumulate of fp32 products in the same K order
    acc = np.zeros((H * W, Co), f32)
    A32 = (A_hi + A_lo).astype(f32) ; B32 = (B_hi + B_lo).astype(f32)
    for g in groups:
        for k in g:
            acc = (acc + A32[:, k:k + 1] * B32[k:k + 1, :]).astype(f32)
    e = np.abs((acc.astype(f64) / (sa * sw)).T.reshape(Co, H, W) - ref)
    out["plain fp32 FMA-less chain (yardstick)"] = (e.mean() / rmax, e.max() / rmax)

    # ---- Winograd F(2x2, 3x3)
    Bt = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], f64)
    G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], f64)
    At = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], f64)
    th, tw = H // 2, W // 2
    # input tiles d[c, ty, tx, 4, 4]; transform in fp32 (every intermediate rounded: two passes of adds)
    d = np.stack([np.stack([xp[:, i:i + 2 * th:2, j:j + 2 * tw:2] for j in range(4)], axis=-1) for i in range(4)], axis=-2)   # [C, th, tw, 4(i), 4(j)]
    t1 = np.einsum("ai,cyxij->cyxaj", Bt, d.astype(f64)).astype(f32)
    V = np.einsum("cyxaj,bj->cyxab", t1.astype(f64), Bt).astype(f32)                                                           # [C, th, tw, 4, 4]
    U = np.einsum("ai,ocij,bj->ocab", G, w.astype(f64), G)                                                                   # fp64 at pack time
    sv, su = scale_for(np.abs(V).max()), scale_for(np.abs(U).max())
    Vh, Vl = split16((V * f32(sv)).astype(f32))
    Uh, Ul = split16((U * su).astype(f32))
    kg = [np.arange(s * 16, s * 16 + 16) for s in range(C // 16)]
    for nm, cg in (("winograd, one chain of 96", 0), ("winograd, chains of 8 k-groups", 8)):
        Mm = np.zeros((th * tw, Co, 4, 4), f32)
        for a in range(4):
            for b in range(4):
                Ah, Al = Vh[:, :, :, a, b].reshape(C, -1).T, Vl[:, :, :, a, b].reshape(C, -1).T
                Bh, Bl = Uh[:, :, a, b].T, Ul[:, :, a, b].T
                Mm[:, :, a, b] = mfma_chain(Ah, Al, Bh, Bl, kg, cg)
        Mm = (Mm.astype(f64) / (sv * su)).astype(f32)                                  # un-scale: exact (powers of two)
        t2 = np.einsum("ai,pnij->pnaj", At, Mm.astype(f64)).astype(f32)
        Y = np.einsum("pnaj,bj->pnab", t2.astype(f64), At).astype(f32)                 # [tiles, Co, 2, 2]
        y = Y.reshape(th, tw, Co, 2, 2).transpose(2, 0, 3, 1, 4).reshape(Co, H, W)
        e = np.abs(y.astype(f64) - ref)
        out[nm] = (e.mean() / rmax, e.max() / rmax)
    # ---- Winograd F(2, 3) along x only (1.5 x fewer products): V = d B per (row, column pair), 4 positions, the three tap ROWS stay a K loop
    B1 = Bt                                                                                                       # 4 x 4, V_p = sum_i Bt[p, i] d_i
    G1 = G                                                                                                        # 4 x 3
    dx = np.stack([xp[:, :, j:j + 2 * tw:2] for j in range(4)], axis=-1)                                           # [C, H+2, tw, 4]
    V1 = np.einsum("pi,cyxi->cyxp", B1, dx.astype(f64)).astype(f32)                                               # one add each: exact in fp64, rounded once
    U1 = np.einsum("pj,ocyj->ocyp", G1, w.astype(f64))                                                            # [Co, C, 3(ky), 4(pos)]
    sv, su = scale_for(np.abs(V1).max()), scale_for(np.abs(U1).max())
    Vh, Vl = split16((V1 * f32(sv)).astype(f32))
    Uh, Ul = split16((U1 * su).astype(f32))
    for nm, cg in (("winograd 1-D (x), chains of 2 slabs", 6), ("winograd 1-D (x), chains of 4 slabs", 12), ("winograd 1-D (x), one chain", 0)):
        M1 = np.zeros((H * tw, Co, 4), f32)
        for p_ in range(4):
            Ah = np.concatenate([Vh[:, ky:ky + H, :, p_].reshape(C, -1).T for ky in range(3)], axis=1)             # [H*tw, 3C]
            Al = np.concatenate([Vl[:, ky:ky + H, :, p_].reshape(C, -1).T for ky in range(3)], axis=1)
            Bh = np.concatenate([Uh[:, :, ky, p_].T for ky in range(3)], axis=0)                                   # [3C, Co]
            Bl = np.concatenate([Ul[:, :, ky, p_].T for ky in range(3)], axis=0)
            g1 = [np.arange(ky * C + s * 16, ky * C + s * 16 + 16) for s in range(C // 16) for ky in range(3)]     # slab-major, 3 tap rows per slab
            M1[:, :, p_] = mfma_chain(Ah, Al, Bh, Bl, g1, cg)
        M1 = (M1.astype(f64) / (sv * su)).astype(f32)
        y0 = ((M1[:, :, 0] + M1[:, :, 1]).astype(f32) + M1[:, :, 2]).astype(f32)
        y1 = ((M1[:, :, 1] - M1[:, :, 2]).astype(f32) - M1[:, :, 3]).astype(f32)
        y = np.stack([y0, y1], axis=-1).reshape(H, tw, Co, 2).transpose(2, 0, 1, 3).reshape(Co, H, W)
        e = np.abs(y.astype(f64) - ref)
        out[nm] = (e.mean() / rmax, e.max() / rmax)
    # ---- Winograd F(4, 3) along x: six products per four outputs = HALF the direct form's, 3/4 of F(2,3)'s.  Interpolation
    # points 0, +-1, +-2, inf (Lavin & Gray): B^T has entries up to 5, A^T up to 8, G down to 1/24 -- the transform's own conditioning is what
    # is being measured.  Same rules as the 1-D form above: input transform on the fp32 activation BEFORE the split (three adds / multiplies
    # per value, each rounded to fp32), filter transform in fp64 at pack time, chains of two slabs, output transform in fp32.
    Bt4 = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], f64)
    G4 = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], f64)
    At4 = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], f64)
    tq = W // 4
    dx4 = np.stack([xp[:, :, j:j + 4 * tq:4] for j in range(6)], axis=-1)                                          # [C, H+2, tq, 6]: columns 4 q - 1 + j
    # every entry of B^T d is a sum of up to four terms: evaluated left to right in fp32 (each partial sum rounded), as a kernel would
    V4 = np.zeros(dx4.shape[:3] + (6,), f32)
    for p_ in range(6):
        acc4 = np.zeros(dx4.shape[:3], f32)
        for i in range(6):
            if Bt4[p_, i]:
                acc4 = (acc4.astype(f64) + Bt4[p_, i] * dx4[..., i].astype(f64)).astype(f32)
        V4[..., p_] = acc4
    U4 = np.einsum("pj,ocyj->ocyp", G4, w.astype(f64))
    sv, su = scale_for(np.abs(V4).max()), scale_for(np.abs(U4).max())
    Vh, Vl = split16((V4 * f32(sv)).astype(f32))
    Uh, Ul = split16((U4 * su).astype(f32))
    for nm, cg in (("winograd F(4,3) along x, chains of 2 slabs", 6), ("winograd F(4,3) along x, one chain", 0)):
        M4 = np.zeros((H * tq, Co, 6), f32)
        for p_ in range(6):
            Ah = np.concatenate([Vh[:, ky:ky + H, :, p_].reshape(C, -1).T for ky in range(3)], axis=1)
            Al = np.concatenate([Vl[:, ky:ky + H, :, p_].reshape(C, -1).T for ky in range(3)], axis=1)
            Bh = np.concatenate([Uh[:, :, ky, p_].T for ky in range(3)], axis=0)
            Bl = np.concatenate([Ul[:, :, ky, p_].T for ky in range(3)], axis=0)
            g1 = [np.arange(ky * C + s * 16, ky * C + s * 16 + 16) for s in range(C // 16) for ky in range(3)]
            M4[:, :, p_] = mfma_chain(Ah, Al, Bh, Bl, g1, cg)
        M4 = (M4.astype(f64) / (sv * su)).astype(f32)
        ys = []
        for o_ in range(4):
            acc4 = np.zeros(M4.shape[:2], f32)
            for p_ in range(6):
                if At4[o_, p_]:
                    acc4 = (acc4.astype(f64) + At4[o_, p_] * M4[:, :, p_].astype(f64)).astype(f32)
            ys.append(acc4)
        y = np.stack(ys, axis=-1).reshape(H, tq, Co, 4).transpose(2, 0, 1, 3).reshape(Co, H, W)
        e = np.abs(y.astype(f64) - ref)
        out[nm] = (e.mean() / rmax, e.max() / rmax)
    out["(operand range: max|V| / max|x|  F(2,3) / F(4,3))"] = (float(np.abs(V1).max() / np.abs(x).max()), float(np.abs(V4).max() / np.abs(x).max()))
    return out


if __name__ == "__main__":
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    rows = {}
    for s in range(seeds):
        for k, v in run(s).items():
            rows.setdefault(k, []).append(v)
    base = np.mean([m for m, _ in rows["direct, chains of 2 slabs (product)"]])
    print(f"{'variant':44s} {'mean|err|/max|ref|':>20s} {'max|err|/max|ref|':>20s}   x direct(mean)")
    for k, v in rows.items():
        if k.startswith("("):
            print(f"{k:44s} {np.mean([a for a, _ in v]):20.3f} {np.mean([b for _, b in v]):20.3f}")
            continue
        m, x = np.mean([a for a, _ in v]), np.max([b for _, b in v])
        print(f"{k:44s} {m:20.3e} {x:20.3e}   {m / base:6.2f}")
