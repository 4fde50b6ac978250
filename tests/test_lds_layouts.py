"""Host-side check of the LDS tile layouts the round-4 kernels rely on (csrc/sdpa_bwd.hip `t3_swz`, csrc/window_attn.hip
`wa_swz`, csrc/gemm_tn.hip's 256 x 128 tile): the per-lane byte addresses of every access pattern are recomputed here from the
formulas in the kernels' comments, and every group of lanes the LDS services in one cycle (MI355X_MICROARCH.md "LDS": ds_read_b128
in 16-lane groups {0-3,12-15,20-27} / {4-11,16-19,28-31} / +32, ds_read_b64(_tr_b16) in 32-lane groups, 64 banks of 4 bytes) must
touch 64 distinct banks -- the property the PMC counter SQ_LDS_BANK_CONFLICT = 0 confirmed on the device."""
import itertools

B128_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
B128_GROUPS += [[l + 32 for l in g] for g in B128_GROUPS[:2]]
B64_GROUPS = [list(range(32)), list(range(32, 64))]


def banks(addr, nbytes):
    return [((addr + 4 * i) // 4) % 64 for i in range(nbytes // 4)]


def conflict_free(addr_of_lane, groups, nbytes):
    for g in groups:
        hit = list(itertools.chain.from_iterable(banks(addr_of_lane(l), nbytes) for l in g))
        if len(set(hit)) != len(hit):
            return False
    return True


def pi_row(i):
    return (i & 0x13) | ((i & 4) << 1) | ((i & 8) >> 1)


def t3_swz(r):
    return (((r >> 1) & 1) << 2) | ((r >> 2) & 3)


def test_attention_backward_tile_image_serves_row_and_transposed_fragments_without_bank_conflicts():
    # image: 64 rows x 128 bytes, chunk c of row r at position c ^ t3_swz(r)
    for blk in range(2):
        for ks in range(4):                                    # row fragments: lane (li, half) reads chunk 2 ks + half of row pi(li)
            def addr(lane):
                li, half = lane & 31, lane >> 5
                r = 32 * blk + pi_row(li)
                return r * 128 + (((2 * ks + half) ^ t3_swz(r)) << 4)
            assert conflict_free(addr, B128_GROUPS, 16), (blk, ks)
        for s2 in range(2):
            for db in range(2):
                for q in range(2):                             # transposed fragments: rows 4 q .. 4 q + 3 of a 16-row group
                    def addr(lane):
                        g, t = lane >> 4, lane & 15
                        r = 32 * blk + 16 * s2 + 8 * (g >> 1) + 4 * q + (t >> 2)
                        c = 4 * db + 2 * (g & 1) + ((t & 3) >> 1)
                        return r * 128 + ((c ^ t3_swz(r)) << 4) + (t & 1) * 8
                    assert conflict_free(addr, B64_GROUPS, 8), (blk, s2, db, q)
    # the round-3 swizzle (gemm.hip g_swz<4>) serves the row fragments but NOT the transposed reads: why t3_swz exists
    old = lambda r: ((r >> 1) & 3) | (((r >> 4) & 1) << 2)
    def addr_old(lane):
        g, t = lane >> 4, lane & 15
        r = 8 * (g >> 1) + (t >> 2)
        c = 2 * (g & 1) + ((t & 3) >> 1)
        return r * 128 + ((c ^ old(r)) << 4) + (t & 1) * 8
    assert not conflict_free(addr_old, B64_GROUPS, 8)


def test_window_attention_backward_images():
    # [token][32 channels] row-major image (64-byte rows), transposing reads of 4 rows x 16 columns per 16-lane group
    for row0 in (0, 4, 8, 16, 36, 56):
        def addr(lane):
            g, t = lane >> 4, lane & 15
            return (row0 + 8 * (g >> 1) + (t >> 2)) * 64 + (16 * (g & 1) + 4 * (t & 3)) * 2
        assert conflict_free(addr, B64_GROUPS, 8), row0
    # [query][key] image (128-byte rows, chunk ^ wa_swz(row) = t3_swz): B fragments by transposing reads ...
    for kq in range(4):
        for jb in range(2):
            for q in range(2):
                def addr(lane):
                    g, t = lane >> 4, lane & 15
                    r = 16 * kq + 8 * (g >> 1) + 4 * q + (t >> 2)
                    chunk = 4 * jb + 2 * (g & 1) + ((t & 3) >> 1)
                    return r * 128 + ((chunk ^ t3_swz(r)) << 4) + 8 * (t & 1)
                assert conflict_free(addr, B64_GROUPS, 8), (kq, jb, q)
    # ... and its 8-byte stores (lane = query row, four consecutive keys): ds_write_b64 is serviced 16 lanes at a time
    for ib in range(2):
        for jb in range(2):
            for g4 in range(4):
                def addr(lane):
                    li, hf = lane & 31, lane >> 5
                    r = ib * 32 + li
                    return r * 128 + (((jb * 4 + g4) ^ t3_swz(r)) << 4) + 8 * hf
                groups = [list(range(16 * k, 16 * k + 16)) for k in range(4)]
                for grp in groups:                             # 16 lanes x 8 bytes: 32 distinct banks
                    hit = list(itertools.chain.from_iterable(banks(addr(l), 8) for l in grp))
                    assert len(set(hit)) == len(hit), (ib, jb, g4)


def test_weight_gradient_wide_tile_transposed_reads():
    # dY tile: 32 token rows x 512 bytes, X tile: 32 x 256 bytes; chunk c of row r at position c ^ ((r & 3) << 2)
    for pitch, nblk in ((512, 8), (256, 4)):
        for blk in range(nblk):
            for sk in range(2):
                for q in range(2):
                    def addr(lane):
                        g, t = lane >> 4, lane & 15
                        r = 16 * sk + 8 * (g >> 1) + 4 * q + (t >> 2)
                        c = 4 * blk + 2 * (g & 1) + ((t & 3) >> 1)
                        return r * pitch + ((c ^ ((r & 3) << 2)) << 4) + (t & 1) * 8
                    assert conflict_free(addr, B64_GROUPS, 8), (pitch, blk, sk, q)
    # without the swizzle the four rows of a group sit 512 (256) bytes apart: the same banks four (two) times
    def plain(lane):
        g, t = lane >> 4, lane & 15
        return (8 * (g >> 1) + (t >> 2)) * 512 + ((2 * (g & 1) + ((t & 3) >> 1)) << 4) + (t & 1) * 8
    assert not conflict_free(plain, B64_GROUPS, 8)
