"""Lane-level model of csrc/linear_mfma.h (CPU, numpy): the staging map, the LDS image,
the fragment reads and the accumulator -> output map of one 128 x 128 block tile are
replayed with an MFMA emulated from the documented gfx950 operand layout
(v_mfma_f32_32x32x16_bf16: lane l holds A[i = l & 31][k = 8 (l >> 5) .. + 7], B[k][j = l & 31];
D[row = (r & 3) + 8 (r >> 2) + 4 (l >> 5)][col = l & 31]) and must reproduce x @ w.T.
Guards the index arithmetic of the kernel against edits; the kernel itself is checked
against torch on the GPU (tests/test_linear_gpu.py)."""
import numpy as np

BM = BN = 128
BK = 32
ROW = 40            # bf16 elements per LDS row (kLinRow)
PLANE = 128 * ROW


def _mfma_32x32x16(a_frag, b_frag, acc):
    """a_frag, b_frag: (64, 8) per-lane operands; acc: (64, 16) per-lane accumulators."""
    A = np.zeros((32, 16))
    B = np.zeros((16, 32))
    for l in range(64):
        A[l & 31, 8 * (l >> 5):8 * (l >> 5) + 8] = a_frag[l]
        B[8 * (l >> 5):8 * (l >> 5) + 8, l & 31] = b_frag[l]
    D = A @ B
    out = acc.copy()
    for l in range(64):
        for r in range(16):
            out[l, r] += D[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5), l & 31]
    return out


def _block_tile(x, w, m0, n0):
    M, K = x.shape
    N = w.shape[0]
    y = np.full((M, N), np.nan)
    acc = np.zeros((4, 2, 2, 64, 16))
    for kc in range(0, K, BK):
        lds = np.zeros(2 * PLANE)            # planes: A, W (hi only: the model checks indices)
        for tid in range(256):
            srow, skq = tid >> 2, (tid & 3) * 8
            for p in range(2):
                gm = min(m0 + p * 64 + srow, M - 1)
                gn = min(n0 + p * 64 + srow, N - 1)
                off = (p * 64 + srow) * ROW + skq
                lds[off:off + 8] = x[gm, kc + skq:kc + skq + 8]
                lds[PLANE + off:PLANE + off + 8] = w[gn, kc + skq:kc + skq + 8]
        for wave in range(4):
            wm, wn = wave >> 1, wave & 1
            for ks in range(2):
                af = np.zeros((2, 64, 8))
                bf = np.zeros((2, 64, 8))
                for lane in range(64):
                    frow, fk = lane & 31, (lane >> 5) * 8
                    for t in range(2):
                        ao = (wm * 64 + frow) * ROW + fk + t * 32 * ROW + ks * 16
                        bo = (wn * 64 + frow) * ROW + fk + t * 32 * ROW + ks * 16
                        af[t, lane] = lds[ao:ao + 8]
                        bf[t, lane] = lds[PLANE + bo:PLANE + bo + 8]
                for i in range(2):
                    for j in range(2):
                        acc[wave, i, j] = _mfma_32x32x16(af[i], bf[j], acc[wave, i, j])
    for wave in range(4):
        wm, wn = wave >> 1, wave & 1
        for lane in range(64):
            for j in range(2):
                n = n0 + wn * 64 + j * 32 + (lane & 31)
                for i in range(2):
                    mb = m0 + wm * 64 + i * 32 + 4 * (lane >> 5)
                    for r in range(16):
                        m = mb + (r & 3) + 8 * (r >> 2)
                        if n < N and m < M:
                            y[m, n] = acc[wave, i, j, lane, r]
    return y


def test_block_tile_indexing_reproduces_gemm():
    rng = np.random.default_rng(0)
    M, N, K = 150, 200, 64            # ragged in both directions: 2 x 2 tiles with tails
    x = rng.standard_normal((M, K))
    w = rng.standard_normal((N, K))
    want = x @ w.T
    got = np.full((M, N), np.nan)
    for m0 in range(0, M, BM):
        for n0 in range(0, N, BN):
            t = _block_tile(x, w, m0, n0)
            mask = ~np.isnan(t)
            assert not (mask & ~np.isnan(got)).any(), "two tiles wrote the same element"
            got[mask] = t[mask]
    assert not np.isnan(got).any(), "an output element was never written"
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-12)


def test_xcd_tile_map_covers_every_tile_once():
    for nbm, nbn in ((1, 1), (7, 2), (8, 3), (313, 2), (1445, 6)):
        grid = ((nbm + 7) // 8) * 8 * nbn
        seen = set()
        for b in range(grid):
            xcd, seq = b & 7, b >> 3
            mt, nt = (seq // nbn) * 8 + xcd, seq % nbn
            if mt < nbm:
                assert (mt, nt) not in seen
                seen.add((mt, nt))
        assert len(seen) == nbm * nbn


def test_lds_image_is_conflict_free_for_b128_fragment_reads():
    # ds_read_b128 service groups (MI355X_MICROARCH.md, LDS): 16 lanes each, 64 banks of 4 bytes
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
              [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    groups += [[l + 32 for l in g] for g in groups]
    for g in groups:
        banks = set()
        for lane in g:
            byte = ((lane & 31) * ROW + (lane >> 5) * 8) * 2
            for d in range(4):
                banks.add((byte // 4 + d) % 64)
        assert len(banks) == 64


# -- packed weight image, transposed-tile epilogue and grouped output (same lane-level style) --------

def _pack_weight(w):
    """lin_pack_weight_kernel's index arithmetic: blob[(n/128 * K/32 + k/32) * 2 * PLANE + plane * PLANE
    + (n % 128) * ROW + k % 32] (rows >= N and the 8-element row pad are zero; one plane modelled)."""
    N, K = w.shape
    nt, kc = (N + 127) // 128, K // 32
    blob = np.zeros(nt * kc * 2 * PLANE)
    for n in range(N):
        for k in range(K):
            blob[((n // 128) * kc + k // 32) * 2 * PLANE + (n % 128) * ROW + k % 32] = w[n, k]
    return blob


def test_packed_weight_chunk_is_the_lds_image_the_fragments_read():
    """WMODE 3: chunk (nt, c) of the blob is copied verbatim into the W area; the B fragment of lane l
    for tile j, k-step ks must be W[n0 + wn*64 + j*32 + (l & 31), kc + ks*16 + 8*(l >> 5) .. + 7]."""
    rng = np.random.default_rng(1)
    N, K = 200, 96
    w = rng.standard_normal((N, K))
    blob = _pack_weight(w)
    kcn = K // 32
    for nt in range(2):
        for c in range(kcn):
            chunk = blob[(nt * kcn + c) * 2 * PLANE:(nt * kcn + c) * 2 * PLANE + PLANE]     # hi plane = LDS image
            for wn in range(2):
                for j in range(2):
                    for ks in range(2):
                        for lane in (0, 5, 31, 32, 47, 63):
                            bo = (wn * 64 + (lane & 31)) * ROW + (lane >> 5) * 8 + j * 32 * ROW + ks * 16
                            n = nt * 128 + wn * 64 + j * 32 + (lane & 31)
                            k = c * 32 + ks * 16 + (lane >> 5) * 8
                            want = w[n, k:k + 8] if n < N else np.zeros(8)
                            np.testing.assert_array_equal(chunk[bo:bo + 8], want)


def test_transposed_tile_epilogue_and_grouped_output_cover_every_element_once():
    """SWAP epilogue: the MFMA computes D[n][m] (W fragment as the A operand), so lane l holds output
    row m = l & 31 and, in registers 4g .. 4g+3, columns nb + 8g .. +3 with nb = 4 (l >> 5): one
    16-byte store per g.  With group_cols the column n lands in matrix n / group_cols."""
    M, N, gc = 150, 512, 256
    groups = N // gc
    seen = np.zeros((groups, M, gc), dtype=int)
    for m0 in range(0, M, BM):
        for n0 in range(0, N, BN):
            grp, ncol0 = n0 // gc, (n0 // gc) * gc
            for wave in range(4):
                wm, wn = wave >> 1, wave & 1
                for lane in range(64):
                    for i in range(2):
                        m = m0 + wm * 64 + i * 32 + (lane & 31)
                        for j in range(2):
                            nb = n0 + wn * 64 + j * 32 + 4 * (lane >> 5)
                            for g in range(4):
                                n = nb + 8 * g
                                if m < M and n < N:
                                    seen[grp, m, n - ncol0:n - ncol0 + 4] += 1
    assert (seen == 1).all()


# -- row-panel kernel (csrc/linear_panel.h): DMA slot map, in-place split, fragment reads, fragment-order weights ------

def _panel_row_of(q, rl):
    return ((rl >> 1) & 3) | ((rl & 1) << 3) | ((q & 1) << 2) | ((q >> 1) << 4)


def _panel_pack_weight(w, n_tiles32):
    """lin_panel_pack_weight_kernel: blob[T][sg][plane][lane][e]; one plane modelled (values, not bf16)."""
    N, K = w.shape
    nstep = K // 16
    blob = np.zeros((n_tiles32, nstep, 64, 8))
    for T in range(n_tiles32):
        for sg in range(nstep):
            for lane in range(64):
                n = T * 32 + (lane & 31)
                j = 2 * (sg % 16) + (lane >> 5)
                p, c = j >> 3, j & 7
                k = (sg // 16) * 256 + (2 * p) * 32 + 4 * c
                if n < N:
                    blob[T, sg, lane, :4] = w[n, k:k + 4]
                    blob[T, sg, lane, 4:] = w[n, k + 32:k + 36]
    return blob


def _panel_block(x, w, m0, MT, NT, NW):
    """One workgroup of linear_panel_kernel<., MT, NT, NW>: returns the (rows, N) outputs it stores (NaN elsewhere)."""
    M, K = x.shape
    N = w.shape[0]
    BM, TW = MT * 32, NT * 32
    NPAIR = (BM // 8) * 4
    nct = (N + TW - 1) // TW
    nstep = K // 16
    blob = _panel_pack_weight(w, ((N + 63) // 64) * 2)
    y = np.full((M, N), np.nan)
    acc = np.zeros((NW * 8, MT, NT, 64, 16))           # [column tile][i][j][lane][reg] (more than enough tiles)
    for half in range(K // 256):
        # LDS as 16-byte granules of 8 values: [pair][hi half (64 slots) | lo half]; the model keeps the 8 values of a
        # lane's two DMA slots together in the hi slot (the split writes hi8 there; lo8 goes to the same slot + 1 KiB)
        lds = np.zeros((NPAIR, 64, 8))
        PPW = NPAIR // NW
        QPW = PPW // 4
        for wave in range(NW):
            for lane in range(64):
                d_rl, d_cc = lane >> 3, lane & 7
                for u in range(QPW):
                    q = wave * QPW + u
                    row = _panel_row_of(q, d_rl)
                    assert 0 <= row < BM
                    c = d_cc ^ (row & 7)
                    gm = min(m0 + row, M - 1)
                    for p in range(4):
                        k = half * 256 + (2 * p) * 32 + c * 4
                        lds[q * 4 + p, lane, :4] = x[gm, k:k + 4]           # DMA slot A: line 2p
                        lds[q * 4 + p, lane, 4:] = x[gm, k + 32:k + 36]     # DMA slot B: line 2p + 1
        for wave in range(NW):
            for ct in range(wave, nct, NW):
                for s in range(16):
                    af = np.zeros((MT, 64, 8))
                    bf = np.zeros((NT, 64, 8))
                    for lane in range(64):
                        f_r, f_h = lane & 31, lane >> 5
                        f_q0 = ((f_r >> 2) & 1) | ((f_r >> 4) << 1)
                        f_rl = ((f_r & 3) << 1) | ((f_r >> 3) & 1)
                        f_x = f_r & 7
                        addr = f_q0 * 4 * 2048 + (f_rl * 8 + (((2 * (s & 3)) + f_h) ^ f_x)) * 16 + (s >> 2) * 2048
                        for i in range(MT):
                            a = addr + i * (4 * 4 * 2048)
                            pair, slot = a // 2048, (a % 2048) // 16
                            assert slot < 64 and a % 16 == 0             # hi half of the pair
                            af[i, lane] = lds[pair, slot]
                        for j in range(NT):
                            bf[j, lane] = blob[ct * NT + j, half * 16 + s, lane]
                    for i in range(MT):
                        for j in range(NT):
                            # D[n][m]: W fragment as the MFMA's A operand
                            acc[ct, i, j] = _mfma_32x32x16(bf[j], af[i], acc[ct, i, j])
    for wave in range(NW):
        for ct in range(wave, nct, NW):
            n0 = ct * TW
            for lane in range(64):
                for i in range(MT):
                    m = m0 + i * 32 + (lane & 31)
                    for j in range(NT):
                        nb = n0 + j * 32 + 4 * (lane >> 5)
                        for g in range(4):
                            n = nb + 8 * g
                            if m < M and n < N:
                                assert np.isnan(y[m, n:n + 4]).all()
                                y[m, n:n + 4] = acc[ct, i, j, lane, 4 * g:4 * g + 4]
    return y


def test_panel_kernel_indexing_reproduces_gemm():
    rng = np.random.default_rng(2)
    for (MT, NT, NW), (M, N, K) in (((2, 2, 4), (150, 200, 256)), ((4, 1, 8), (150, 328, 256)),
                                    ((2, 2, 4), (70, 192, 512)), ((4, 1, 8), (130, 256, 512))):
        x = rng.standard_normal((M, K))
        w = rng.standard_normal((N, K))
        want = x @ w.T
        got = np.full((M, N), np.nan)
        BM = MT * 32
        for m0 in range(0, M, BM):
            t = _panel_block(x, w, m0, MT, NT, NW)
            mask = ~np.isnan(t)
            assert not (mask & ~np.isnan(got)).any(), "two workgroups wrote the same element"
            got[mask] = t[mask]
        assert not np.isnan(got).any(), "an output element was never written"
        np.testing.assert_allclose(got, want, rtol=1e-11, atol=1e-11)


def test_panel_dma_slots_cover_the_panel_once_in_full_lines():
    for MT, NW in ((2, 4), (4, 8)):
        BM = MT * 32
        NPAIR = (BM // 8) * 4
        QPW = NPAIR // NW // 4
        seen = set()
        for wave in range(NW):
            for u in range(QPW):
                q = wave * QPW + u
                for p in range(4):
                    lines = {}
                    for lane in range(64):
                        row = _panel_row_of(q, lane >> 3)
                        c = (lane & 7) ^ (row & 7)
                        assert (row, p, c) not in seen
                        seen.add((row, p, c))
                        lines.setdefault(row, set()).add(c)
                    # one DMA instruction = 8 rows x one full 128-byte line (8 lanes of 16 bytes per row)
                    assert len(lines) == 8 and all(v == set(range(8)) for v in lines.values())
        assert len(seen) == BM * 4 * 8


def test_panel_lds_image_is_conflict_free_for_b128_fragment_reads():
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
              [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    groups += [[l + 32 for l in g] for g in groups]
    for s in range(16):
        for g in groups:
            banks = set()
            for lane in g:
                f_r, f_h = lane & 31, lane >> 5
                f_q0 = ((f_r >> 2) & 1) | ((f_r >> 4) << 1)
                f_rl = ((f_r & 3) << 1) | ((f_r >> 3) & 1)
                byte = f_q0 * 4 * 2048 + (f_rl * 8 + (((2 * (s & 3)) + f_h) ^ (f_r & 7))) * 16 + (s >> 2) * 2048
                for d in range(4):
                    banks.add((byte // 4 + d) % 64)
            assert len(banks) == 64, (s, g)


def test_chain_kernel_plane_writes_feed_the_next_stage():
    """csrc/linear_chain.h: a stage's output tile set (8 wavefronts x 64 x 32 in MFMA accumulator layout: lane holds
    row i * 32 + (lane & 31), columns 32 w + 4 (lane >> 5) + 8 g + e in register 4 g + e) is written straight into the
    plane buffer (``p_addr``); the next stage's fragment reads (``f_addr``, as in the row-panel kernel) and the
    fragment-order weight image must then reproduce x @ w.T — i.e. the register -> slot map is the one the DMA + split
    pass of linear_panel.h produces."""
    rng = np.random.default_rng(3)
    BM, NW = 64, 8
    x = rng.standard_normal((BM, 256))
    w = rng.standard_normal((256, 256))
    lds = np.full(32 * 2048 // 2, np.nan)                      # hi planes only, one entry per bf16 element (2 bytes)
    for wave in range(NW):
        for lane in range(64):
            f_r, f_h = lane & 31, lane >> 5
            f_q0 = ((f_r >> 2) & 1) | ((f_r >> 4) << 1)
            f_rl = ((f_r & 3) << 1) | ((f_r >> 3) & 1)
            f_x = f_r & 7
            for i in range(2):
                for g in range(4):
                    byte = (f_q0 * 4 + (wave >> 1)) * 2048 + (f_rl * 8 + ((f_h + 2 * g) ^ f_x)) * 16 + (wave & 1) * 8 \
                        + i * (4 * 4 * 2048)
                    assert byte % 8 == 0 and (byte % 2048) < 1024       # hi half of the pair
                    row, col = i * 32 + f_r, 32 * wave + 4 * f_h + 8 * g
                    assert np.isnan(lds[byte // 2:byte // 2 + 4]).all()
                    lds[byte // 2:byte // 2 + 4] = x[row, col:col + 4]
    blob = _panel_pack_weight(w, 8)
    got = np.zeros((BM, 256))
    for wave in range(NW):
        acc = np.zeros((2, 64, 16))
        for s in range(16):
            af = np.zeros((2, 64, 8))
            for lane in range(64):
                f_r, f_h = lane & 31, lane >> 5
                f_q0 = ((f_r >> 2) & 1) | ((f_r >> 4) << 1)
                f_rl = ((f_r & 3) << 1) | ((f_r >> 3) & 1)
                addr = f_q0 * 4 * 2048 + (f_rl * 8 + (((2 * (s & 3)) + f_h) ^ (f_r & 7))) * 16 + (s >> 2) * 2048
                for i in range(2):
                    a = addr + i * (4 * 4 * 2048)
                    af[i, lane] = lds[a // 2:a // 2 + 8]
            assert not np.isnan(af).any()
            for i in range(2):
                acc[i] = _mfma_32x32x16(blob[wave, s], af[i], acc[i])
        for lane in range(64):
            for i in range(2):
                for r in range(16):
                    got[i * 32 + (lane & 31), 32 * wave + 4 * (lane >> 5) + 8 * (r >> 2) + (r & 3)] = acc[i, lane, r]
    np.testing.assert_allclose(got, x @ w.T, rtol=1e-11, atol=1e-11)


# ------------------------------------------------------------------------------------------------------------------
# csrc/wgrad_tr.h: bf16 planes + ds_read_b64_tr_b16 fragments of the weight-gradient kernel (round 4)

def _tr_read(addr_of_lane, lds_u16):
    """gfx950 ``ds_read_b64_tr_b16`` as measured by tools/probes/tr_read_probe.hip (profiles/r4/r4h_*_probe.txt): in
    every 16-lane group, lane i receives as element j the element (i & 3) of the 8-byte piece addressed by lane
    (i >> 2) + 4 j of the same group."""
    out = []
    for lane in range(64):
        base = lane & ~15
        i = lane & 15
        vals = []
        for j in range(4):
            a = addr_of_lane[base + (i >> 2) + 4 * j]
            assert a % 8 == 0
            vals.append(lds_u16[a // 2 + (i & 3)])
        out.append(vals)
    return out


def test_transposing_read_model_reproduces_the_hardware_probe():
    import os
    import re
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r4",
                        "r4h_ds_read_b64_tr_b16_probe.txt")
    text = open(path).read()
    lds = list(range(8192))
    patterns = {0: lambda l: l * 8,
                1: lambda l: (l & 3) * 8 + ((l >> 2) & 3) * 64 + (l >> 4) * 256,
                2: lambda l: (l & 15) * 64 + (l >> 4) * 8,
                3: lambda l: 0}
    for p, fn in patterns.items():
        block = text.split(f"pattern {p}\n")[1].split("pattern ")[0]
        got = {int(m.group(1)): [int(v) for v in m.group(2).split()]
               for m in re.finditer(r"lane\s+(\d+):((?:\s+\d+){4})", block)}
        assert len(got) == 64
        want = _tr_read([fn(l) for l in range(64)], lds)
        for lane in range(64):
            assert got[lane] == want[lane], (p, lane, got[lane], want[lane])


def test_weight_gradient_planes_and_transposed_fragments():
    """Replay of wgrad_tr.h's address arithmetic: the split-and-store pass writes element (row m, column n) of a chunk
    tile at plane byte m * 320 + 2 n; lane L of a wavefront (wn / wk, column tile t, k-step ks) must receive rows
    16 ks + 8 (L >> 5) + 0..7 of column 64 w + 32 t + (L & 31) — the MFMA operand layout — and the 32 pieces a half-wave
    addresses in one transposing read must lie on 32 distinct bank pairs (row stride 80 banks = 16 mod 64)."""
    ROWB, STRIDE = 320, 160
    plane = [0] * (32 * STRIDE)
    for m in range(32):                      # writer: thread (prow0 = tid >> 5, c4 = 4 (tid & 31)) rows prow0 + 8 i
        for n in range(128):
            plane[(m * ROWB + 2 * n) // 2] = m * 1000 + n
    for w in range(2):
        for t in range(2):
            for ks in range(2):
                for second in range(2):
                    addr = []
                    for lane in range(64):
                        s, g0, g1 = lane & 15, (lane >> 4) & 1, lane >> 5
                        frow, fcolb = 8 * g1 + (s >> 2), (16 * g0 + 4 * (s & 3)) * 2
                        addr.append(frow * ROWB + (w * 64 + t * 32) * 2 + fcolb + ks * 16 * ROWB + second * 4 * ROWB)
                    got = _tr_read(addr, plane)
                    for lane in range(64):
                        for j in range(4):
                            m = 16 * ks + 8 * (lane >> 5) + 4 * second + j
                            n = 64 * w + 32 * t + (lane & 31)
                            assert got[lane][j] == m * 1000 + n, (w, t, ks, second, lane, j)
                    for half in range(2):        # 32 lanes x 8 bytes: 32 distinct pairs of 4-byte banks (64 banks)
                        banks = [(addr[half * 32 + l] // 4) % 64 for l in range(32)]
                        assert len(set(banks)) == 32 and all(b % 2 == 0 for b in banks), (w, t, ks, second, half)
    # writer: a wavefront's 64 threads store 8 bytes each: lanes 0..31 one row (256 contiguous bytes), lanes 32..63 the next
    for wave in range(4):
        for half in range(2):
            tids = [wave * 64 + half * 32 + l for l in range(32)]
            a = [((t_ >> 5) * ROWB + (t_ & 31) * 8) for t_ in tids]
            assert len({(x // 4) % 64 for x in a}) == 32
