"""The operand planes of the ray-MLP chain (dense.hip) on a machine without a GPU: the kernels' index arithmetic lives in
csrc/dense_layout.h, which libsixdgs_hostcheck.so instantiates for the host.  A tag (ray, feature, plane) is walked through

    epilogue of layer L (accumulator register -> feature -> bytes in HBM)  ->  loader of layer L + 1 (bytes in HBM -> LDS image)
    ->  MFMA operand fragment (lane, k-step -> 8 consecutive inputs of one ray)

for both activation layouts and both tile shapes: every input of every ray must arrive in the fragment slot the MFMA expects, rays beyond a
ragged tile's end must alias the last valid ray, and the permuted weight rows must make a lane's registers whole 16-byte chunks."""
import ctypes as C
import importlib

import numpy as np
import pytest


@pytest.fixture(scope="module")
def hc():
    b = importlib.import_module("6dgs_amd.build")
    lib = C.CDLL(b.build_hostcheck())
    lib.hc_dl_plane_offset.restype = C.c_longlong
    lib.hc_dl_plane_offset.argtypes = [C.c_int, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_int]
    for f in ("hc_dl_lds_offset", "hc_dl_frag_offset", "hc_dl_load_ray", "hc_dl_load_chunk8", "hc_dl_cm_src_offset"):
        getattr(lib, f).restype = C.c_uint
    return lib


def test_constants_and_row_permutation(hc):
    slab_b, prow, gran, gran_slab, run = (hc.hc_dl_const(i) for i in range(5))
    assert (slab_b, gran, gran_slab, run) == (128, 128, 128 * 128, 128 * 16) and prow == 144
    perm = [hc.hc_dl_row_perm(m) for m in range(32)]
    assert sorted(perm) == list(range(32)) and [perm[p] for p in perm] == list(range(32))          # an involution on the 32 rows of a block
    for lane in (0, 31, 32, 63):
        h = lane >> 5
        rows = [hc.hc_dl_acc_row(lane, r) for r in range(16)]
        assert rows == [8 * (r >> 2) + 4 * h + (r & 3) for r in range(16)]                       # the MFMA 32x32 result layout
        feats = [hc.hc_dl_acc_feature(1, lane, r) for r in range(16)]
        for p in range(2):                                                                         # register groups 2p, 2p+1: one whole chunk
            assert feats[8 * p:8 * p + 8] == list(range(16 * p + 8 * h, 16 * p + 8 * h + 8))
        assert [hc.hc_dl_acc_feature(0, lane, r) for r in range(16)] == rows


def _produce(hc, cm, ntm, ntn, n, m_rays):
    """Epilogue of a layer with n features over m_rays rays: tags at the byte offsets the kernel stores to (one tag per fp16 element)."""
    fp, rt, nslab = 128 * ntm, 64 * ntn, n // 32
    hbm = np.full(((m_rays + 127) // 128) * 128 * nslab * 64, -1, np.int64)
    for tile in range((m_rays + rt - 1) // rt):
        for ps in range(n // fp):
            f0 = ps * fp
            for wave in range(8):
                wm, wn = wave >> 1, wave & 1
                for lane in range(64):
                    for tm in range(ntm):
                        for tn in range(ntn):
                            gr = tile * rt + wn * 32 * ntn + (lane & 31) + 32 * tn
                            if gr >= m_rays:
                                continue
                            slab = (f0 >> 5) + tm * 4 + wm
                            for r in range(16):                      # after the split: register r's value as one fp16 of plane h and one of plane l
                                feat_in_slab = hc.hc_dl_acc_feature(int(cm), lane, r)
                                c, e = feat_in_slab >> 3, feat_in_slab & 7
                                for pl in range(2):
                                    off = hc.hc_dl_plane_offset(int(cm), gr, nslab, slab, pl, c) + 2 * e
                                    assert hbm[off // 2] == -1
                                    hbm[off // 2] = (gr * n + slab * 32 + feat_in_slab) * 2 + pl
    return hbm


def _consume(hc, cm, hbm, ntm, ntn, k_in, m_rays, slabs):
    fp, rt, ks, kwl = 128 * ntm, 64 * ntn, k_in // 32, 2 * ntm
    prow = hc.hc_dl_const(1)
    for tile in range((m_rays + rt - 1) // rt):
        r0 = tile * rt
        lrmax = min(rt - 1, m_rays - 1 - r0)
        for s in slabs:
            lds = np.full(512 * prow // 2, -1, np.int64)
            for tid in range(512):
                for jp in range(8 - kwl):
                    ray = min(hc.hc_dl_load_ray(int(cm), tid, jp), lrmax)
                    c8 = hc.hc_dl_load_chunk8(int(cm), tid)
                    if cm:
                        src = (r0 >> 7) * ks * 16384 + s * 16384 + hc.hc_dl_cm_src_offset(ray, c8, ks * 16384)
                    else:
                        src = hc.hc_dl_plane_offset(0, r0 + ray, ks, s, c8 >> 2, c8 & 3)
                    assert src == hc.hc_dl_plane_offset(int(cm), r0 + ray, ks, s, c8 >> 2, c8 & 3)
                    dst = hc.hc_dl_lds_offset(fp + hc.hc_dl_load_ray(int(cm), tid, jp), c8 >> 2, c8 & 3)
                    assert lds[dst // 2] == -1                                    # every 16-byte cell of the ray rows written exactly once
                    lds[dst // 2:dst // 2 + 8] = hbm[src // 2:src // 2 + 8]
            for wave in range(8):
                wn = wave & 1
                for lane in range(64):
                    for kstep in range(2):
                        for pl in range(2):
                            for t in range(ntn):
                                row0 = fp + wn * 32 * ntn + 32 * t
                                a = hc.hc_dl_frag_offset(row0, lane, kstep, pl)
                                ray = r0 + min(wn * 32 * ntn + 32 * t + (lane & 31), lrmax)
                                want = [((ray * k_in + s * 32 + kstep * 16 + (lane >> 5) * 8 + e) * 2 + pl) for e in range(8)]
                                assert lds[a // 2:a // 2 + 8].tolist() == want, (tile, s, wave, lane, kstep, pl, t)


@pytest.mark.parametrize("cm", [True, False])
@pytest.mark.parametrize("producer,n", [((2, 4), 512), ((3, 2), 384)])
def test_a_layers_output_arrives_in_the_next_layers_fragments(hc, cm, producer, n):
    m_rays = 300                                                                   # two granules and a ragged third; a ragged tile in both shapes
    hbm = _produce(hc, cm, producer[0], producer[1], n, m_rays)
    nslab = n // 32
    for gr in (0, 127, 128, 299):                                                  # every element of a valid ray written, at its own position
        for feat in range(n):
            for pl in range(2):
                off = hc.hc_dl_plane_offset(int(cm), gr, nslab, feat // 32, pl, (feat % 32) // 8) + 2 * (feat % 8)
                assert hbm[off // 2] == (gr * n + feat) * 2 + pl
    for consumer in ((2, 4), (3, 2)):
        _consume(hc, cm, hbm, consumer[0], consumer[1], n, m_rays, slabs=(0, nslab // 2, nslab - 1))
