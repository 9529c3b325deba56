"""A moving camera through rtx_scene_set_view, which since round 5 queues its work (source copies of the prune records, first-frame cost
estimate, tile lists written on the device) without waiting for the device: every view of a sequence must
give the picture a freshly loaded scene with that camera gives -- and the oracle's (Scene::render, scene.cpp:595-606) -- bit for bit; coming
back to an earlier view must give its picture again (stale source copies or lists would not)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def frame(g, w, h, mode=-1):
    fb = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
    mask = torch.zeros((h, w), dtype=torch.uint8, device="cuda")
    g.set_frame_mode(mode)
    g.render_frame(fb, mask)
    assert g.frame_status() == 0
    return fb.cpu().numpy(), mask.cpu().numpy()


POSES = [((0.0, 0.0, 0.0), (0.0, 0.0, 0.0)), ((0.15, 0.1, 0.2), (-4.0, 9.0, 2.0)), ((-0.3, 0.25, 0.4), (3.0, -14.0, -1.0)), ((0.0, 0.6, -0.5), (-25.0, 0.0, 0.0))]


@pytest.mark.parametrize("name,w,h", [("cfg2_smooth_25k", 320, 200), ("cfg2_smooth_4k", 200, 152), ("mixed_materials", 200, 152)])
def test_views_of_a_sequence_equal_fresh_scenes_and_the_oracle(ra, oracle, tmp_path, name, w, h):
    src = open("scenes/%s.scene" % name).read()
    g = ra.Scene("scenes/%s.scene" % name, w, h)
    g.gpu()
    g.set_knob("verify_lists", 1)
    first = {}
    for rounds in range(2):                      # the second time round every view comes back
        for k, (pos, rot) in enumerate(POSES):
            g.set_camera(pos, rot)
            for mode in (0, 1):
                got, gm = frame(g, w, h, mode)
                if k not in first:
                    # the same camera in the scene file: a fresh GPU scene and the oracle
                    lines = src.split("\n")
                    i = lines.index("[options]")
                    j = next(k2 for k2 in range(i + 1, len(lines)) if lines[k2].startswith("["))      # (lights have a position= of their own)
                    lines[i + 1:j] = [l for l in lines[i + 1:j] if not l.startswith(("position=", "rotation="))] + ["position=%g,%g,%g" % pos, "rotation=%g,%g,%g" % rot]
                    p = tmp_path / ("%s_%d.scene" % (name, k))
                    p.write_text("\n".join(lines))
                    o = oracle.OracleScene(str(p), w, h)
                    ref = o.ssaa(o.pass1())
                    fresh = ra.Scene(str(p), w, h)
                    ff, fm = frame(fresh, w, h, 0)
                    fresh.close()
                    first[k] = (ref, ff, fm)
                ref, ff, fm = first[k]
                assert np.array_equal(bits(got), bits(ff)) and np.array_equal(gm, fm), "view %d (round %d, mode %d) differs from a freshly loaded scene" % (k, rounds, mode)
                d = (bits(got) != bits(ref)).any(-1)
                d[0, :] = False; d[:, 0] = False          # (the reference's uninitialised mask border)
                assert not d.any(), "view %d (round %d, mode %d): %d pixels differ from the oracle" % (k, rounds, mode, int(d.sum()))
    g.close()


def test_set_view_does_not_wait_for_the_device(ra):
    """The host time of a new view must not include the frames still queued on the device."""
    import time
    w = h = 2048
    g = ra.Scene("scenes/cfg2_smooth_250k.scene", w, h)
    fb = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
    mask = torch.zeros((h, w), dtype=torch.uint8, device="cuda")
    for _ in range(3):
        g.render_frame(fb, mask)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); g.render_frame(fb, mask); torch.cuda.synchronize(); one = time.perf_counter() - t0
    pos, rot = g.camera_pose()
    # twelve frames queued, then a new view: setting it must return long before they are done
    for _ in range(12):
        g.render_frame(fb, mask)
    t0 = time.perf_counter()
    g.set_camera(pos + np.float32([0.01, 0, 0]), rot)
    g.gpu()
    dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    assert g.frame_status() == 0
    assert dt < 6 * one, "rtx_scene_set_view took %.3f ms behind twelve queued frames of %.3f ms each: it waited for the device" % (dt * 1e3, one * 1e3)
    g.close()


def test_views_on_a_non_blocking_render_stream(ra):
    """The same camera sequence rendered on a NON-BLOCKING stream (torch.cuda.Stream(): the null stream's implicit ordering does not cover it): the work
    rtx_scene_set_view queues on the null stream must be ordered behind the frames still running on the render stream and before the next one (two events:
    prepBegin / prepEnd / renderOn in rtx_api.hip).  No host synchronisation between a frame and the next view; pictures == the null-stream ones."""
    w, h = 640, 400
    g = ra.Scene("scenes/cfg2_smooth_25k.scene", w, h)
    want = []
    for pos, rot in POSES:
        g.set_camera(pos, rot)
        want.append(frame(g, w, h, 0))
    side = torch.cuda.Stream()
    fbs = [torch.zeros((h, w, 3), dtype=torch.float32, device="cuda") for _ in POSES]
    masks = [torch.zeros((h, w), dtype=torch.uint8, device="cuda") for _ in POSES]
    torch.cuda.synchronize()
    g.set_frame_mode(0)
    for rounds in range(3):
        for k, (pos, rot) in enumerate(POSES):
            g.set_camera(pos, rot)                       # queued behind the previous view's frame, which may still be running on `side`
            g.render_frame(fbs[k], masks[k], stream=side)
        side.synchronize()
        assert g.frame_status() == 0
        for k in range(len(POSES)):
            assert np.array_equal(bits(fbs[k].cpu().numpy()), bits(want[k][0])) and np.array_equal(masks[k].cpu().numpy(), want[k][1]), "view %d, round %d" % (k, rounds)
    g.close()


@pytest.mark.parametrize("how", ["rows", "ownership"])
def test_list_cache_misses_on_a_non_blocking_render_stream(ra, how):
    """a tile list that is NOT prepared by rtx_scene_set_view -- a row sub-range, or any view under row ownership (prepareView returns early) -- is
    written by rtxTileListKernel inside the render call.  That kernel must run on the caller's stream: on the null stream nothing orders it against a
    non-blocking render stream, neither before the launch that reads the list nor behind the earlier launch still reading the recycled entry.  A moving
    camera, more views than the list cache holds, no host synchronisation inside a round; pictures == the ones rendered on the null stream."""
    w, h = 640, 400
    g = ra.Scene("scenes/cfg2_smooth_25k.scene", w, h)
    poses = [((0.02 * k, 0.01 * k, 0.0), (0.0, 1.5 * k, 0.0)) for k in range(20)]      # 20 views x (1 or 2 lists) > the 16 cached lists
    if how == "ownership":
        g.set_row_ownership(64, 2, 1, True)
    rows = (96, 304) if how == "rows" else None
    want = []
    for pos, rot in poses:
        g.set_camera(pos, rot)
        fb = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
        mask = torch.zeros((h, w), dtype=torch.uint8, device="cuda")
        g.set_frame_mode(0)
        g.render_frame(fb, mask, rows=rows)
        torch.cuda.synchronize()
        assert g.frame_status() == 0
        want.append((fb.cpu().numpy(), mask.cpu().numpy()))
    side = torch.cuda.Stream()
    fbs = [torch.zeros((h, w, 3), dtype=torch.float32, device="cuda") for _ in poses]
    masks = [torch.zeros((h, w), dtype=torch.uint8, device="cuda") for _ in poses]
    torch.cuda.synchronize()
    for mode in (0, 1):
        g.set_frame_mode(mode)
        for k, (pos, rot) in enumerate(poses):
            g.set_camera(pos, rot)
            fbs[k].zero_(); masks[k].zero_()
            side.wait_stream(torch.cuda.current_stream())
            g.render_frame(fbs[k], masks[k], rows=rows, stream=side)
        side.synchronize()
        assert g.frame_status() == 0
        for k in range(len(poses)):
            assert np.array_equal(bits(fbs[k].cpu().numpy()), bits(want[k][0])) and np.array_equal(masks[k].cpu().numpy(), want[k][1]), "%s: view %d, mode %d" % (how, k, mode)
    g.close()
