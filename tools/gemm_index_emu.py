"""CPU emulation of the data movement of gemm_ntd / gemm_nt6 (clipa_amd/csrc/gemm_nt.hip): DMA piece placement with
the source-side chunk swizzle and the B-row permutation -> LDS image -> 16x16x32 fragment reads -> MFMA -> direct
epilogue mapping.  Exact integer arithmetic; run at design time (no GPU):  python tools/gemm_index_emu.py
"""
import numpy as np


def bperm(r64):
    bj, i = r64 >> 4, r64 & 15
    return 32 * (bj >> 1) + 8 * (i >> 2) + 4 * (bj & 1) + (i & 3)


def mfma_16x16x32(wfrag, xfrag):
    """wfrag/xfrag: [64 lanes][8]. lane (c=l&15, g=l>>4) supplies row c, k = 8g..8g+7.
    returns out[64 lanes][4]: lane (c, g) holds D[i=4g+r][j=c] = sum_k W[i][k] X[j][k]."""
    W = np.zeros((16, 32), np.int64)
    X = np.zeros((16, 32), np.int64)
    for l in range(64):
        c, g = l & 15, l >> 4
        W[c, 8 * g:8 * g + 8] = wfrag[l]
        X[c, 8 * g:8 * g + 8] = xfrag[l]
    D = W @ X.T
    out = np.zeros((64, 4), np.int64)
    for l in range(64):
        c, g = l & 15, l >> 4
        out[l] = D[4 * g:4 * g + 4, c]
    return out


def emulate(kind, A, B):
    """A [256, K], B [BNt, K] (K = one K step) -> C [256, BNt] via the kernel's index maps."""
    if kind == "nt6":
        nwaves, BNt, KS, rowbytes = 4, 128, 32, 64
    else:
        nwaves, BNt, KS, rowbytes = 8, 256, 64, 128
    assert A.shape == (256, KS) and B.shape == (BNt, KS)
    ldsA = np.full((256 * KS,), -10**9, np.int64)     # element-granular LDS images (bf16 elements)
    ldsB = np.full((BNt * KS,), -10**9, np.int64)
    for wave in range(nwaves):
        for lane in range(64):
            if kind == "nt6":
                r16 = lane >> 2
                chunk = (lane & 3) ^ ((0 - (lane >> 4)) & 3)
                rowA0, rowB0 = 16 * wave + r16, bperm(16 * wave + r16)
                for j in range(4):     # A piece wave + 4j at LDS byte offset (wave + 4j) * 1024 + lane * 16
                    dst = ((wave + 4 * j) * 1024 + lane * 16) // 2
                    ldsA[dst:dst + 8] = A[rowA0 + 64 * j, chunk * 8:chunk * 8 + 8]
                for j in range(2):
                    dst = ((wave + 4 * j) * 1024 + lane * 16) // 2
                    ldsB[dst:dst + 8] = B[rowB0 + 64 * j, chunk * 8:chunk * 8 + 8]
            else:
                r8 = lane >> 3
                chunk = (lane & 7) ^ (((lane >> 4) + 4 * (wave & 1)) & 7)
                i16, bjw = 8 * (wave & 1) + r8, wave >> 1
                rowA0, rowB0 = wave * 8 + r8, bperm(16 * bjw + i16)
                for j in range(4):     # piece j*8 + wave at byte offset wave*1024 + j*8192 + lane*16
                    dst = (wave * 1024 + j * 8192 + lane * 16) // 2
                    ldsA[dst:dst + 8] = A[rowA0 + 64 * j, chunk * 8:chunk * 8 + 8]
                    ldsB[dst:dst + 8] = B[rowB0 + 64 * j, chunk * 8:chunk * 8 + 8]
    assert (ldsA > -10**9).all() and (ldsB > -10**9).all(), "LDS image not fully covered"
    C = np.zeros((256, BNt), np.int64)
    for wave in range(nwaves):
        wm, wn = (wave >> 1, wave & 1) if kind == "nt6" else (wave >> 2, wave & 3)
        acc = np.zeros((4, 8, 64, 4), np.int64)
        for kk in range(KS // 32):
            fa = np.zeros((8, 64, 8), np.int64)
            fb = np.zeros((4, 64, 8), np.int64)
            for lane in range(64):
                l15, g4 = lane & 15, lane >> 4
                if kind == "nt6":
                    coff = (g4 ^ ((0 - (l15 >> 2)) & 3)) << 4
                    for ai in range(8):
                        o = ((wm * 128 + l15) * 64 + coff + ai * 1024) // 2
                        fa[ai, lane] = ldsA[o:o + 8]
                    for bj in range(4):
                        o = ((wn * 64 + l15) * 64 + coff + bj * 1024) // 2
                        fb[bj, lane] = ldsB[o:o + 8]
                else:
                    sw16 = (l15 >> 1) & 7
                    coff = ((4 * kk + g4) ^ sw16) << 4
                    for ai in range(8):
                        o = ((wm * 128 + l15) * 128 + ai * 2048 + coff) // 2
                        fa[ai, lane] = ldsA[o:o + 8]
                    for bj in range(4):
                        o = ((wn * 64 + l15) * 128 + bj * 2048 + coff) // 2
                        fb[bj, lane] = ldsB[o:o + 8]
            for bj in range(4):
                for ai in range(8):
                    acc[bj, ai] += mfma_16x16x32(fb[bj], fa[ai])
        # direct epilogue: acc[bj][ai][r] of lane (c, g) = C[wm*128 + 16 ai + c][wn*64 + 32 (bj>>1) + 8 g + 4 (bj&1) + r]
        for lane in range(64):
            c, g = lane & 15, lane >> 4
            for s in range(2):
                n = wn * 64 + 32 * s + 8 * g
                for ai in range(8):
                    m = wm * 128 + 16 * ai + c
                    C[m, n:n + 4] = acc[2 * s, ai, lane]
                    C[m, n + 4:n + 8] = acc[2 * s + 1, ai, lane]
    return C


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    for kind, BNt, KS in (("ntd", 256, 64), ("nt6", 128, 32)):
        A = rng.integers(-8, 8, (256, KS))
        B = rng.integers(-8, 8, (BNt, KS))
        C = emulate(kind, A, B)
        ok = np.array_equal(C, A @ B.T)
        print(kind, "index maps", "OK" if ok else "WRONG")
        assert ok
