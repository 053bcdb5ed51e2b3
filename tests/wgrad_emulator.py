"""Lane-level CPU emulation of ``nastar_conv3x3_wgrad_kernel`` + ``nastar_wgrad_reduce_kernel`` (csrc/nastar_conv_wgrad.hip.h).

TEST INFRASTRUCTURE.  Mirrors the kernel statement by statement: chunking into whole image rows, the staging plan (dz rows, framed
activation rows with zeros outside the image), the padded LDS row strides, the per-lane fragment addresses, the MEASURED semantics of
gfx950's ``ds_read_b64_tr_b16`` (tools/ubench/tr16.hip: in every 16-lane group lane i receives element i & 3 of the 8-byte rows addressed
by lanes (i >> 2) + 4 j), the MFMA operand / accumulator lane layout, the split-precision products, the per-split partial tiles and
their fixed-order reduction into torch's [co][ci][3][3] layout.  Tiny shapes only (pure numpy)."""
import numpy as np

WG_MAX_PIX, WG_MAX_KS = 96, 6


def chunk_rows(H, W):
    if W < 2 or W > WG_MAX_PIX or H <= 0:
        return 0
    if 64 % W == 0 and H % (64 // W) == 0:
        return 64 // W
    for r in range(WG_MAX_PIX // W, 0, -1):
        if H % r == 0:
            return r
    return 0


def chunk_images(H, W):
    if H * W > 48 or chunk_rows(H, W) < H:
        return 1
    return max(1, min(WG_MAX_PIX // (H * W), 200 // ((H + 2) * (W + 2))))


def row_bytes(r):
    return r + 64 if r % 128 == 0 else r


def tr_read(lds16, addr):
    """ds_read_b64_tr_b16 for one wavefront: addr[64] byte addresses of each lane's 8-byte row -> [64][4] elements"""
    out = np.zeros((64, 4), np.float64)
    for lane in range(64):
        base, i = lane & ~15, lane & 15
        for j in range(4):
            src = base + (i >> 2) + 4 * j           # the lane whose row is read
            out[lane, j] = lds16[addr[src] // 2 + (i & 3)]
    return out


def run(dz, a, B, H, W, CO, CI, co_real, ci_real, split, nsplit=3, out_scale=1.0):
    """dz [B*H*W][M*CO], a [B*H*W][M*CI] flat np.float16 (M = 2: [hi | lo]) -> dw [co_real][ci_real][3][3] float32"""
    M = 2 if split else 1
    COB = 2 if CO % 64 == 0 else 1
    CIB = 2 if CI % 64 == 0 else 1
    NTHR = 192 * COB * CIB
    R = chunk_rows(H, W)
    assert R > 0
    G = chunk_images(H, W)
    NP, PW = G * R * W, W + 2
    BP, BS = R * W, (R + 2) * (W + 2)
    KS = (NP + 15) // 16
    ZROWS = KS * 16
    RDZ, RA = row_bytes(COB * 64 * M), row_bytes(CIB * 64 * M)
    CPZ, CPA = M * COB * 4, M * CIB * 4
    nslot_a = G * BS
    sdz, sa = M * CO, M * CI
    nchunk = (B + G - 1) // G if G > 1 else B * H // R
    nsplit = max(1, min(nsplit, nchunk))
    rows_per_img = H // R
    ntco, ntci = CO // (32 * COB), CI // (32 * CIB)
    part = np.zeros((nsplit, 9, CI, CO), np.float64)
    lds_bytes = ZROWS * RDZ + nslot_a * RA
    lane = np.arange(64)
    r16, grp, kh = lane & 15, (lane >> 4) & 1, lane >> 5
    chan = 16 * grp + 4 * (r16 & 3)
    for blk in range(nsplit * ntco * ntci):
        split_id = blk % nsplit
        t = blk // nsplit
        tco, tci = t % ntco, t // ntco
        co0, ci0 = tco * 32 * COB, tci * 32 * CIB
        acc = np.zeros((3 * COB * CIB, 3, 32, 32), np.float64)  # [wave][dx][row = co][col = ci]
        for ch in range(split_id, nchunk, nsplit):
            if G > 1:
                b, y0 = ch * G, 0
                gcount = min(G, B - b)
            else:
                b, y0, gcount = ch // rows_per_img, (ch % rows_per_img) * R, 1
            p0 = (b * H + y0) * W
            lds = np.full(lds_bytes // 2, 777.0, np.float16)
            lds[NP * RDZ // 2: ZROWS * RDZ // 2] = 0  # rows [NP, ZROWS) of the dz tile are zeroed once
            for q in range(NP * CPZ):                 # dz staging plan
                pix, c = q // CPZ, q % CPZ
                half, cc = c // (COB * 4), c % (COB * 4)
                src = (p0 + pix) * sdz + half * CO + co0 + cc * 8
                dst = (pix * RDZ + half * (COB * 64) + cc * 16) // 2
                lds[dst:dst + 8] = dz[src:src + 8] if pix // BP < gcount else 0
            for q in range(nslot_a * CPA):            # framed activation rows
                slot, c = q // CPA, q % CPA
                half, cc = c // (CIB * 4), c % (CIB * 4)
                gi, sb = slot // BS, slot % BS
                sr, sc = sb // PW, sb % PW
                dst = (ZROWS * RDZ + slot * RA + half * (CIB * 64) + cc * 16) // 2
                y = y0 + sr - 1
                if 1 <= sc <= W and 0 <= y < H and gi < gcount:
                    src = (p0 + gi * BP + (sr - 1) * W + (sc - 1)) * sa + half * CI + ci0 + cc * 8
                    lds[dst:dst + 8] = a[src:src + 8]
                else:
                    lds[dst:dst + 8] = 0
            for wave in range(3 * COB * CIB):
                wdy, wco, wci = wave % 3, (wave // 3) % COB, wave // (3 * COB)
                adz0 = (8 * kh + (r16 >> 2)) * RDZ + wco * 64 + chan * 2
                for ks in range(KS):
                    def frag(a0, a1):
                        f = np.concatenate((tr_read(lds, a0), tr_read(lds, a1)), axis=1)  # [64][8]: pixels +0..3, +4..7
                        assert not np.any(f == 777.0), "fragment read an LDS element nobody staged"
                        return f
                    z0 = adz0 + 16 * ks * RDZ
                    zh = frag(z0, z0 + 4 * RDZ)
                    zl = frag(z0 + COB * 64, z0 + 4 * RDZ + COB * 64) if split else None
                    aa = []
                    for tt in range(2):
                        pix = 16 * ks + 8 * kh + 4 * tt + (r16 >> 2)
                        pc = np.minimum(pix, NP - 1)
                        gi, pb = pc // BP, pc % BP
                        row, col = pb // W, pb % W
                        aa.append(ZROWS * RDZ + (gi * BS + (row + 1 + (wdy - 1)) * PW + col + 1) * RA + wci * 64 + chan * 2)
                    for dx in range(3):
                        toff = (dx - 1) * RA
                        xh = frag(aa[0] + toff, aa[1] + toff)
                        # MFMA 32x32x16: A row m = lane % 32, k = 8 (lane / 32) + e; B col n = lane % 32, same k
                        def mfma(A, Bm):
                            return np.einsum("hre,hce->rc", A.reshape(2, 32, 8), Bm.reshape(2, 32, 8))
                        acc[wave, dx] += mfma(zh, xh)
                        if split:
                            xl = frag(aa[0] + toff + CIB * 64, aa[1] + toff + CIB * 64)
                            acc[wave, dx] += mfma(zl, xh) + mfma(zh, xl)
        for wave in range(3 * COB * CIB):
            wdy, wco, wci = wave % 3, (wave // 3) % COB, wave // (3 * COB)
            for dx in range(3):
                tap = wdy * 3 + dx
                # D[row = co][col = ci] -> part[split][tap][ci][co]
                part[split_id, tap, ci0 + wci * 32: ci0 + wci * 32 + 32, co0 + wco * 32: co0 + wco * 32 + 32] = acc[wave, dx].T
    s = part.sum(axis=0) * out_scale                 # [tap][ci][co] (the kernel sums the splits in a fixed 4-way interleaved order)
    return s.transpose(2, 1, 0)[:co_real, :ci_real].reshape(co_real, ci_real, 3, 3).astype(np.float32)
