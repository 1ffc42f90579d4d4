#!/usr/bin/env python3
"""Index-arithmetic model of tsgemm_s8_ares_kernel (csrc/tsgemm.hip), run on the HOST: every address expression of the
kernel -- DMA source / LDS destination with the XOR swizzle, the (chunk, k-step) sequence over the three weight stages,
the 16-byte fragment reads, the v_mfma_i32_32x32x32_i8 operand / accumulator layout the validated kernels rely on, the
epilogue's (row, column) of acc[g][4 rq + e] and its beyond-the-buffer predication -- transcribed per thread and
executed with numpy against a @ w.T.  It checks the arithmetic of the staged kernel, not the kernel (no device here):
waits, barriers and register allocation are outside its reach.  usage: tsgemm_ares_model.py [M N K]"""
import sys

import numpy as np

kTsBN, kTsG, kThreads, kStages = 256, 5, 512, 3
kTsW, kTsX = kTsBN * 128, kTsG * 32 * 128
kOob = 0xFFFFFF00


def swz8(r):
    return (r ^ (r >> 3)) & 7


def run(M, N, K, n_blocks=3, seed=0):
    rng = np.random.default_rng(seed)
    a = rng.integers(-127, 128, (M, K), dtype=np.int8)
    w = rng.integers(-127, 128, (N, K), dtype=np.int8)
    res = rng.integers(-127, 128, (M, N), dtype=np.int8)
    a_flat, w_flat = a.reshape(-1).view(np.uint8), w.reshape(-1).view(np.uint8)
    NK = K // 128
    nchunk, T, chunk_bytes = N // kTsBN, (N // kTsBN) * NK, kTsBN * K
    units = (M + 31) // 32
    out = np.full((M, N), -99999, np.int64)
    res_seen = np.zeros((M, N), np.int64)
    tid = np.arange(kThreads)
    lane, wave = tid & 63, tid >> 6
    prow, pchunk, hi = lane >> 3, lane & 7, lane >> 5
    fa = wave * 32 + (lane & 31)

    def load16(flat, nbytes, voff, soff):
        """buffer_load ... lds: 16 bytes per lane from voff + soff, zeros when voff is beyond the buffer"""
        o = np.zeros((kThreads, 16), np.uint8)
        ok = voff < nbytes
        idx = (voff[ok] + soff)[:, None] + np.arange(16)[None, :]
        o[ok] = flat[idx]
        return o

    for bi in range(n_blocks):
        per, extra = units // n_blocks, units % n_blocks
        u_begin = bi * per + min(bi, extra)
        u_end = u_begin + per + (1 if bi < extra else 0)
        lds_a = np.zeros(2 * kTsX, np.uint8)
        lds_w = np.zeros(kStages * kTsW, np.uint8)
        w_off = [((wave * 4 + j) * 8 + prow) * K + ((pchunk ^ swz8((wave * 4 + j) * 8 + prow)) << 4) for j in range(4)]

        def dma_w(t, buf):
            c, s = divmod(t, NK)
            soff = c * chunk_bytes + s * 128
            for j in range(4):
                dst = buf * kTsW + wave * 4096 + j * 1024 + lane * 16          # lane-linear within the piece
                data = load16(w_flat, N * K, w_off[j], soff)
                lds_w[dst[:, None] + np.arange(16)[None, :]] = data

        for u0 in range(u_begin, u_end, kTsG):
            G = min(kTsG, u_end - u0)
            r0, pieces_x = u0 * 32, G * 4
            for s in range(NK):
                for j in range(3):
                    act = wave + 8 * j < pieces_x
                    row = (wave + 8 * j) * 8 + prow
                    off = (r0 + row) * K + ((pchunk ^ swz8(row)) << 4)
                    data = load16(a_flat, M * K, np.where(act, off, kOob), s * 128)
                    dst = s * kTsX + (wave + 8 * j) * 1024 + lane * 16
                    sel = act
                    lds_a[dst[sel][:, None] + np.arange(16)[None, :]] = data[sel]
            dma_w(0, 0)
            if T > 1:
                dma_w(1, 1)
            t = 0
            for c in range(nchunk):
                acc = np.zeros((G, kThreads, 16), np.int64)
                for s in range(NK):
                    if t + 2 < T:
                        assert (t + 2) % kStages != t % kStages and (t + 2) % kStages != (t + 1) % kStages
                        dma_w(t + 2, (t + 2) % kStages)
                    Wb, Xb = (t % kStages) * kTsW, s * kTsX
                    for ks in range(4):
                        cc = 2 * ks + hi
                        a_frag = lds_w[(Wb + fa * 128 + ((cc ^ swz8(fa)) << 4))[:, None] + np.arange(16)[None, :]].view(np.int8)
                        for g in range(G):
                            xr = g * 32 + (lane & 31)
                            b_frag = lds_a[(Xb + xr * 128 + ((cc ^ swz8(xr)) << 4))[:, None] + np.arange(16)[None, :]].view(np.int8)
                            # v_mfma_i32_32x32x32_i8 per wave: A row i = lane & 31 (k half = lane >> 5), B row j likewise;
                            # D[i][j] += sum_k A[i][k] B[j][k]; lane L holds D[8 rq + 4 (L >> 5) + e][L & 31] in acc[4 rq + e]
                            for wv in range(8):
                                sl = slice(wv * 64, wv * 64 + 64)
                                A = np.zeros((32, 32), np.int64)
                                B = np.zeros((32, 32), np.int64)
                                for L in range(64):
                                    A[L & 31, (L >> 5) * 16:(L >> 5) * 16 + 16] = a_frag[sl][L]
                                    B[L & 31, (L >> 5) * 16:(L >> 5) * 16 + 16] = b_frag[sl][L]
                                D = A @ B.T
                                for L in range(64):
                                    for rq in range(4):
                                        for e in range(4):
                                            acc[g, wv * 64 + L, 4 * rq + e] += D[8 * rq + 4 * (L >> 5) + e, L & 31]
                    t += 1
                colb = c * kTsBN + wave * 32 + 4 * hi
                for rq in range(4):
                    col = colb + 8 * rq
                    for g in range(G):
                        m = r0 + g * 32 + (lane & 31)
                        row_o = np.where(m < M, m * N, kOob)
                        off = np.where(row_o == kOob, kOob, row_o + col)
                        ok = off < M * N
                        for e in range(4):
                            mm, nn = np.divmod(off[ok] + e, N)
                            out[mm, nn] = acc[g, ok, 4 * rq + e]
                            res_seen[mm, nn] = res.reshape(-1)[off[ok] + e]
    want = a.astype(np.int64) @ w.astype(np.int64).T
    assert np.array_equal(out, want), (np.argwhere(out != want)[:5], (out != want).mean())
    assert np.array_equal(res_seen, res.astype(np.int64))
    return True


if __name__ == "__main__":
    shapes = [tuple(int(v) for v in sys.argv[1:4])] if len(sys.argv) >= 4 else [(333, 512, 256), (161, 256, 128), (520, 768, 256)]
    for M, N, K in shapes:
        print(M, N, K, "ok" if run(M, N, K) else "MISMATCH", flush=True)


def run_f16(M, N, K, stages, units_per_tile, n_blocks=3, seed=0):
    """tsgemm_f16_ares_kernel<NK, STAGES, UNITS>: the same model with 2-byte elements (small integers, exact in fp16 /
    fp32), tiles of `units_per_tile` units dealt out per block, `stages` weight stages."""
    rng = np.random.default_rng(seed)
    a = rng.integers(-4, 5, (M, K)).astype(np.float16)
    w = rng.integers(-4, 5, (N, K)).astype(np.float16)
    a_flat, w_flat = a.reshape(-1).view(np.uint8), w.reshape(-1).view(np.uint8)
    KB = K * 2
    NK = KB // 128
    UN, kXs = units_per_tile, units_per_tile * 32 * 128
    nchunk, T, chunk_bytes = N // kTsBN, (N // kTsBN) * NK, kTsBN * KB
    units = (M + 31) // 32
    tiles = (units + UN - 1) // UN
    out = np.full((M, N), -99999.0)
    tid = np.arange(kThreads)
    lane, wave = tid & 63, tid >> 6
    prow, pchunk, hi = lane >> 3, lane & 7, lane >> 5
    fa = wave * 32 + (lane & 31)

    def load16(flat, nbytes, voff, soff):
        o = np.zeros((kThreads, 16), np.uint8)
        ok = voff < nbytes
        idx = (voff[ok] + soff)[:, None] + np.arange(16)[None, :]
        o[ok] = flat[idx]
        return o

    for bi in range(n_blocks):
        per, extra = tiles // n_blocks, tiles % n_blocks
        t_begin = bi * per + min(bi, extra)
        t_end = t_begin + per + (1 if bi < extra else 0)
        lds_a = np.zeros(4 * kXs, np.uint8)
        lds_w = np.zeros(stages * kTsW, np.uint8)
        w_off = [((wave * 4 + j) * 8 + prow) * KB + ((pchunk ^ swz8((wave * 4 + j) * 8 + prow)) << 4) for j in range(4)]

        def dma_w(t, buf):
            c, s = divmod(t, NK)
            soff = c * chunk_bytes + s * 128
            for j in range(4):
                dst = buf * kTsW + wave * 4096 + j * 1024 + lane * 16
                lds_w[dst[:, None] + np.arange(16)[None, :]] = load16(w_flat, N * KB, w_off[j], soff)

        for tile_i in range(t_begin, t_end):
            r0 = tile_i * UN * 32
            for s in range(NK):
                for j in range((UN * 4 + 7) // 8):
                    act = wave + 8 * j < UN * 4
                    row = (wave + 8 * j) * 8 + prow
                    off = np.where(r0 + row < M, (r0 + row) * KB + ((pchunk ^ swz8(row)) << 4), kOob)
                    data = load16(a_flat, M * KB, np.where(act, off, kOob), s * 128)
                    dst = s * kXs + (wave + 8 * j) * 1024 + lane * 16
                    lds_a[dst[act][:, None] + np.arange(16)[None, :]] = data[act]
            dma_w(0, 0)
            if stages == 3 and T > 1:
                dma_w(1, 1)
            t = 0
            for c in range(nchunk):
                acc = np.zeros((UN, kThreads, 16))
                for s in range(NK):
                    if t + stages - 1 < T:
                        nxt = (t + stages - 1) % stages
                        assert nxt != t % stages
                        dma_w(t + stages - 1, nxt)
                    Wb, Xb = (t % stages) * kTsW, s * kXs
                    for ks in range(4):
                        cc = 2 * ks + hi
                        a_frag = lds_w[(Wb + fa * 128 + ((cc ^ swz8(fa)) << 4))[:, None] + np.arange(16)[None, :]].view(np.float16)
                        for g in range(UN):
                            xr = g * 32 + (lane & 31)
                            b_frag = lds_a[(Xb + xr * 128 + ((cc ^ swz8(xr)) << 4))[:, None] + np.arange(16)[None, :]].view(np.float16)
                            for wv in range(8):
                                sl = slice(wv * 64, wv * 64 + 64)
                                A, B = np.zeros((32, 16)), np.zeros((32, 16))
                                for L in range(64):   # v_mfma_f32_32x32x16_f16: lane L supplies row L & 31, k half L >> 5 (8 values)
                                    A[L & 31, (L >> 5) * 8:(L >> 5) * 8 + 8] = a_frag[sl][L]
                                    B[L & 31, (L >> 5) * 8:(L >> 5) * 8 + 8] = b_frag[sl][L]
                                D = A @ B.T
                                for L in range(64):
                                    for rq in range(4):
                                        for e in range(4):
                                            acc[g, wv * 64 + L, 4 * rq + e] += D[8 * rq + 4 * (L >> 5) + e, L & 31]
                    t += 1
                colb = c * kTsBN + wave * 32 + 4 * hi
                for rq in range(4):
                    col = colb + 8 * rq
                    for g in range(UN):
                        m = r0 + g * 32 + (lane & 31)
                        row_b = np.where(m < M, m * N * 2, kOob)
                        off = np.where(row_b == kOob, kOob, row_b + col * 2)
                        ok = off < M * N * 2
                        for e in range(4):
                            mm, nn = np.divmod(off[ok] // 2 + e, N)
                            out[mm, nn] = acc[g, ok, 4 * rq + e]
    want = a.astype(np.float64) @ w.astype(np.float64).T
    assert np.array_equal(out, want), (np.argwhere(out != want)[:5], (out != want).mean())
    return True


if __name__ == "__main__" and len(sys.argv) < 4:
    for M, N, K, st, un in [(333, 512, 256, 3, 3), (333, 512, 256, 2, 5), (200, 256, 128, 3, 3), (520, 768, 128, 2, 5)]:
        print("f16", M, N, K, f"stages={st} units={un}", "ok" if run_f16(M, N, K, st, un) else "MISMATCH", flush=True)
