"""GPU parity of the consecutive-frame plane matcher (device part): cape_match_consecutive through the
C ABI against oracle/match_oracle.py fed with the CPU oracle's own per-frame results.  Everything is integer except
the two gates, so the bar is equality."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _oracle_frame(orc, depth):
    import match_oracle

    r = orc.run(depth)
    roots = r.planes[:, 19].astype(int) if len(r.planes) else np.zeros(0, int)
    is_out = np.zeros(len(r.merge_labels), bool)
    is_out[roots] = True
    masks, mroots = match_oracle.plane_masks(r.plane_labels, r.segments, r.merge_labels, is_out)
    assert list(mroots) == list(roots)
    return {"masks": masks, "normals": r.planes[:, 0:3], "d": r.planes[:, 3]}


def _check_batch(ex, orc, frames, flags=0):
    import cape_amd
    import match_oracle

    n = ex.extract_host(frames)
    ex.match_consecutive(n, flags)
    got = ex.matches(n)
    per = [_oracle_frame(orc, f) for f in frames]
    matched = 0
    assert got["n_prev"][0] == 0 and got["n_cur"][0] == len(per[0]["masks"]) and (got["match"][0] == -1).all()
    for f in range(1, n):
        m, ap, ac, inter = match_oracle.match_frame(per[f - 1], per[f], advanced=bool(flags & cape_amd.MATCH_ADVANCED),
                                                    allow_index0=bool(flags & cape_amd.MATCH_ALLOW_INDEX0))
        g = got[f]
        assert g["n_prev"] == len(ap) and g["n_cur"] == len(ac), f"frame {f}: plane counts"
        assert list(g["area_prev"][: len(ap)]) == ap and list(g["area_cur"][: len(ac)]) == ac, f"frame {f}: areas"
        assert np.array_equal(g["inter"][: len(ap), : len(ac)], inter), f"frame {f}: inter areas"
        assert list(g["match"][: len(ap)]) == m, f"frame {f}: matches {list(g['match'][:len(ap)])} != {m}"
        assert (g["match"][len(ap):] == -1).all() and (g["inter"][len(ap):] == 0).all()
        matched += sum(1 for x in m if x >= 0)
    return matched


@pytest.mark.parametrize("scene,seed", [("room", 0), ("room", 5), ("tumlike", 2)])
@pytest.mark.parametrize("flags", [0, 2, 3])
def test_match_consecutive_parity(oracle_mod, scene, seed, flags):
    from cape_amd import Extractor, synth

    gen = getattr(synth, scene)
    intr = synth.TUM_FR1_INTRINSICS if scene == "tumlike" else synth.DEFAULT_INTRINSICS
    frames = np.stack([gen(seed=seed, frame=4 * t) for t in range(10)])
    orc = oracle_mod.Oracle(640, 480, cylinders=False, **intr)
    ex = Extractor(640, 480, cylinders=False, max_batch=len(frames), **intr)
    matched = _check_batch(ex, orc, frames, flags)
    ex.close()
    if flags & 2:
        assert matched >= 9, "a slowly panning camera must keep matching its planes"


def test_match_quirk_index0_never_returned(oracle_mod):
    """map_primitive.cpp:146 `if (selectedIndex <= 0) return` : detected plane 0 is never matched unless the caller
    asks for the corrected behaviour."""
    from cape_amd import Extractor, synth

    frames = np.stack([synth.room(seed=1, frame=t) for t in range(6)])
    ex = Extractor(640, 480, max_batch=6, **synth.DEFAULT_INTRINSICS)
    n = ex.extract_host(frames)
    ex.match_consecutive(n, 0)
    quirk = ex.matches(n)
    ex.match_consecutive(n, 2)
    fixed = ex.matches(n)
    assert not (quirk["match"] == 0).any()
    assert (fixed["match"] == 0).any()
    ex.close()


def test_match_scene_cut_and_merged_planes(oracle_mod):
    """Unrelated consecutive frames (no overlap gates pass by luck only) and faceted frames whose output planes are
    merge groups of several segments."""
    from cape_amd import Extractor, synth

    frames = np.stack([synth.facets(seed=11, frame=0), synth.facets(seed=11, frame=1), synth.facets(seed=22, frame=0),
                       synth.room(seed=0, frame=0), synth.facets(seed=22, frame=1), np.zeros((480, 640), np.float32),
                       synth.room(seed=0, frame=1)])
    orc = oracle_mod.Oracle(640, 480, cylinders=True, **synth.DEFAULT_INTRINSICS)
    ex = Extractor(640, 480, cylinders=True, max_batch=len(frames), **synth.DEFAULT_INTRINSICS)
    for flags in (0, 1, 2, 3):
        _check_batch(ex, orc, frames, flags)
    ex.close()


def test_match_1280x960(oracle_mod):
    from cape_amd import Extractor, synth

    intr = {k: 2 * v for k, v in synth.DEFAULT_INTRINSICS.items()}
    frames = np.stack([synth.room(seed=2, frame=3 * t, width=1280, height=960) for t in range(4)])
    orc = oracle_mod.Oracle(1280, 960, cylinders=True, **intr)
    ex = Extractor(1280, 960, cylinders=True, max_batch=4, **intr)
    assert _check_batch(ex, orc, frames, 2) >= 3
    ex.close()


def test_match_argument_checks():
    from cape_amd import CapeError, Extractor, synth

    ex = Extractor(640, 480, max_batch=4, **synth.DEFAULT_INTRINSICS)
    with pytest.raises(CapeError):
        ex.matches(1)                      # nothing matched yet
    ex.extract_host(np.stack([synth.room(seed=0, frame=0)] * 2))
    with pytest.raises(CapeError):
        ex.match_consecutive(3)            # more frames than the last batch
    with pytest.raises(CapeError):
        ex.match_consecutive(2, flags=8)   # unknown flag
    ex.match_consecutive(2)
    m = ex.matches(2)
    assert m["n_prev"][1] == m["n_cur"][0]
    ex.close()


def test_extract_rejects_misaligned_device_pointers():
    """The streaming kernel loads four pixels per lane: a device pointer that is not aligned to four pixels is an
    argument error, not a fault."""
    import torch
    from cape_amd import CapeError, Extractor, synth

    ex = Extractor(640, 480, max_batch=1, **synth.DEFAULT_INTRINSICS)
    buf = torch.zeros(640 * 480 + 8, dtype=torch.float32, device="cuda")
    with pytest.raises(CapeError):
        ex.extract_device(buf.data_ptr() + 4, 1)
    raw = torch.zeros(640 * 480 + 8, dtype=torch.int16, device="cuda")
    with pytest.raises(CapeError):
        ex.extract_device_u16(raw.data_ptr() + 2, 0.2, 1)
    ex.extract_device(buf.data_ptr() + 16, 1)      # aligned offsets are fine
    ex.extract_device_u16(raw.data_ptr() + 8, 0.2, 1)
    torch.cuda.synchronize()
    ex.close()
