"""Lane-level CPU emulation of ``nastar_conv3x3_flat_kernel`` (neural-astar_amd/csrc/nastar_conv_flat.hip.h).

TEST INFRASTRUCTURE: it mirrors the kernel's index arithmetic statement by statement -- dispatch-id -> (tile, channel block), the flat
halo staging with the zero slot and the rotated 16-byte chunks, the per-lane fragment addresses (incl. the ``^ 32`` k-step trick), the
weight chunk order, the MFMA operand / accumulator lane layout, the split-precision segments and the epilogue -- so that the addressing
logic is checked on the CPU (against torch's conv2d) before the kernel ever runs on a GPU.  Slow (pure numpy per workgroup): tiny shapes.
"""
import numpy as np

FC_TP, FC_KS, FC_PIXB, FC_THREADS, FC_NTQ = 256, 32, 64, 256, 8


def slot_off(slot, c):
    return FC_PIXB + slot * FC_PIXB + (((c + (slot >> 2)) & 3) << 4)


def run(inp, inp2, wpack, scale, shift, B, H, W, C1, C2, COUT, relu, final, ups, split, final_mul=1.0):
    """inp / inp2 / wpack: flat np.float16 arrays in the kernel's layouts; returns out (flat float16 [npix*COUT*(2 if split)]) or the
    fp32 [npix] cost map when ``final``."""
    NT = 32 if (final or COUT % 64) else 64
    NB = NT // 32
    NWC = 9 * 2 * 2 * NT
    NWQ = (NWC + FC_THREADS - 1) // FC_THREADS
    npix = B * H * W
    ntiles = (npix + FC_TP - 1) // FC_TP
    halo = W + 1
    nslot = FC_TP + 2 * halo
    assert nslot * 4 <= FC_NTQ * FC_THREADS
    CIN = C1 + C2
    NSL = CIN // FC_KS
    NSLICE = 3 * NSL if split else NSL
    CINV = 3 * CIN if split else CIN
    st1 = 2 * C1 if split else C1
    st2 = 2 * C2 if split else C2
    HW = H * W
    mult = 2 if split else 1
    out = np.zeros(npix * COUT * mult, np.float16)
    out32 = np.zeros(npix, np.float32)
    lds_bytes = FC_PIXB + nslot * FC_PIXB + NWC * 16
    grid = ((ntiles + 7) // 8) * 8 * (COUT // NT)
    nblk_total = COUT // NT
    tid = np.arange(FC_THREADS)
    for bid in range(grid):
        xcd, j = bid & 7, bid >> 3
        tile = (j // nblk_total) * 8 + xcd
        nblk = j % nblk_total
        if tile >= ntiles:
            continue
        p0, n0 = tile * FC_TP, nblk * NT
        q0 = p0 - halo
        smem = np.zeros(lds_bytes // 2, np.float16)  # halves; byte address / 2
        smem[:] = np.float16(777.0)  # poison: whatever is read must have been written
        smem[:32] = 0  # the zero slot
        wl = FC_PIXB + nslot * FC_PIXB  # byte offset of the weights
        # staging plan
        src1 = np.full((FC_NTQ, FC_THREADS), -1, np.int64)
        src2 = np.full((FC_NTQ, FC_THREADS), -1, np.int64)
        for i in range(FC_NTQ):
            idx = tid + i * FC_THREADS
            c, slot = idx & 3, idx >> 2
            q = q0 + slot
            ok = (slot < nslot) & (q >= 0) & (q < npix)
            if ups:
                b = q // HW
                r = q - b * HW
                y = r // W
                x = r - y * W
                o1 = ((b * (H >> 1) + (y >> 1)) * (W >> 1) + (x >> 1)) * (st1 >> 3) + c  # 16-byte units
            else:
                o1 = q * (st1 >> 3) + c
            o2 = q * (st2 >> 3) + c
            src1[i] = np.where(ok, o1, -1)
            src2[i] = np.where(ok, o2, -1)
        wsrc = np.full((NWQ, FC_THREADS), -1, np.int64)
        for i in range(NWQ):
            q = tid + i * FC_THREADS
            n = q % NT
            r = q // NT
            h = r & 1
            r = r >> 1
            kk = r & 1
            r = r >> 1
            wsrc[i] = np.where(q < NWC, ((r * (CINV >> 3) + kk * 2 + h) * COUT + n0 + n) * 8, -1)

        def stage(s):
            seg = s // NSL if split else 0
            ch = ((s - seg * NSL) if split else s) * FC_KS
            first = ch < C1
            if first:
                base, arr, srcs = ch + (C1 if seg == 1 else 0), inp, src1
            else:
                base, arr, srcs = (ch - C1) + (C2 if seg == 1 else 0), inp2, src2
            for i in range(FC_NTQ):
                for t in range(FC_THREADS):
                    idx = t + i * FC_THREADS
                    c, slot = idx & 3, idx >> 2
                    if slot >= nslot:
                        continue
                    o = srcs[i, t] * 8
                    v = arr[base + o: base + o + 8] if o >= 0 else np.zeros(8, np.float16)
                    a = slot_off(slot, c) // 2
                    smem[a:a + 8] = v
            wb = s * (FC_KS // 8) * COUT * 8
            for i in range(NWQ):
                for t in range(FC_THREADS):
                    q = t + i * FC_THREADS
                    if q >= NWC:
                        continue
                    a = (wl + q * 16) // 2
                    smem[a:a + 8] = wpack[wb + wsrc[i, t]: wb + wsrc[i, t] + 8]

        # read plan
        lane = np.arange(64)
        px, kh = lane & 31, lane >> 5
        baddr = np.zeros((4, 9, 2, 64), np.int64)
        for wave in range(4):
            for pb in range(2):
                lp = wave * 64 + pb * 32 + px
                p = p0 + lp
                r = p % HW
                y = r // W
                x = r - y * W
                for tap in range(9):
                    dy, dx = tap // 3 - 1, tap % 3 - 1
                    ok = (p < npix) & (y + dy >= 0) & (y + dy < H) & (x + dx >= 0) & (x + dx < W)
                    slot = lp + halo + dy * W + dx
                    good = np.array([slot_off(int(sl), int(k)) for sl, k in zip(slot, kh)])
                    baddr[wave, tap, pb] = np.where(ok, good, kh << 4)
        acc = np.zeros((4, 2, NB, 32, 32), np.float64)  # [wave][pb][n][row = channel][col = pixel]
        for s in range(NSLICE):
            stage(s)
            for wave in range(4):
                for kk in range(2):
                    for tap in range(9):
                        wa = []
                        for n in range(NB):
                            addr = wl + ((((tap * 2 + kk) * 2 + kh) * NT) + n * 32 + px) * 16
                            wa.append(np.stack([smem[a // 2: a // 2 + 8] for a in addr]).astype(np.float64))
                        for pb in range(2):
                            addr = baddr[wave, tap, pb] ^ (kk << 5)
                            xb = np.stack([smem[a // 2: a // 2 + 8] for a in addr]).astype(np.float64)
                            assert not np.any(xb == 777.0), "fragment read an LDS chunk nobody staged"
                            b2 = xb.reshape(2, 32, 8)
                            for n in range(NB):
                                a2 = wa[n].reshape(2, 32, 8)
                                acc[wave, pb, n] += np.einsum("hre,hce->rc", a2, b2)
        sc, sh = scale[n0:n0 + NT].astype(np.float64), shift[n0:n0 + NT].astype(np.float64)
        CPP, PPR = NT // 8, 64 // (NT // 8)  # 16-byte chunks per pixel and precision half; pixels per store round
        for wave in range(4):
            for pb in range(2):
                if final:
                    for ln in range(64):
                        p = p0 + wave * 64 + pb * 32 + (ln & 31)
                        if p < npix and (ln >> 5) == 0 and nblk == 0:
                            z = acc[wave, pb, 0, 0, ln & 31] * sc[0] + sh[0]
                            out32[p] = final_mul / (1.0 + np.exp(-z))
                    continue
                for part in range(mult):
                    # the wave-private LDS patch of the kernel's epilogue: 32 pixels x NT fp16, 16-byte chunk index XORed with the pixel pair
                    patch = np.full((32 * NT,), np.float16(777.0), np.float16)
                    for ln in range(64):
                        pxl, khl = ln & 31, ln >> 5
                        for n in range(NB):
                            for g in range(4):
                                cl = n * 32 + 8 * g + 4 * khl
                                chunk = n * 4 + g
                                base = pxl * (NT * 2) + ((chunk ^ ((pxl >> 1) & (CPP - 1))) << 4) + khl * 8  # byte offset
                                for e in range(4):
                                    reg = 4 * g + e
                                    row = (reg & 3) + 8 * (reg >> 2) + 4 * khl
                                    v = acc[wave, pb, n, row, pxl] * sc[cl + e] + sh[cl + e]
                                    if relu:
                                        v = max(v, 0.0)
                                    v = np.float32(min(max(v, -65504.0), 65504.0))
                                    hi = np.float16(v)
                                    patch[base // 2 + e] = hi if part == 0 else np.float16(v - np.float32(hi))
                    assert not np.any(patch == np.float16(777.0)), "epilogue patch has a hole"
                    for j in range(32 // PPR):
                        for ln in range(64):
                            q, chunk = j * PPR + ln // CPP, ln % CPP
                            src = q * (NT * 2) + ((chunk ^ ((q >> 1) & (CPP - 1))) << 4)
                            p = p0 + wave * 64 + pb * 32 + q
                            if p < npix:
                                dst = p * COUT * mult + part * COUT + n0 + chunk * 8
                                out[dst:dst + 8] = patch[src // 2:src // 2 + 8]
    return out32 if final else out
