"""What the product actually runs: rtx_render_frame in the modes and with the band heights that
bench.py / Scene::render() use at full size, the frame kernel's recovery when its single launch gives up, and the
regressions of ADVICE r2 (strip marker above 32768 rows, mask rows under row ownership).  Through the C ABI."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SPLIT, FUSED, AUTO = 0, 1, -1


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available()
    return torch


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def stages(torch, g, band=0, parts=1, part=0):
    H, W = g.height, g.width
    fb = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    mask = torch.zeros((H, W), dtype=torch.uint8, device="cuda")
    g.set_row_ownership(band if parts > 1 else 0, parts, part, True)
    g.render_pass1(fb)
    g.sobel(fb, mask)
    g.render_ssaa(mask, fb)
    torch.cuda.synchronize()
    return fb, mask


def same(torch, a, b):
    return torch.equal(a.view(torch.int32), b.view(torch.int32))


@pytest.mark.parametrize("size", [4096, 8192])
def test_render_frame_fused_and_auto_at_full_size(ra, torch_cuda, size):
    """The 250k-triangle scene at 4096^2 (headline) and 8192^2 (cfg5) through rtx_render_frame: the single launch, forced (cold, then
    warm frames that split slow tiles), gives the three stages' framebuffer and mask bit for bit; left to choose, frames of this size
    take three launches BY RULE (more than 65 536 tiles: the single launch has never won there and is no longer probed -- round 4),
    over 70 frames, with the same pixels.  (The measured choice itself: tests/test_gpu_frame.py at the sizes where it applies.)"""
    from rendering_amd import assets
    torch = torch_cuda
    assets.ensure(["bumpy_250k.obj"])
    g = ra.Scene("scenes/cfg2_smooth_250k.scene", size, size)
    ref_fb, ref_mask = stages(torch, g)
    fb = torch.zeros_like(ref_fb); mask = torch.zeros_like(ref_mask)
    g.set_frame_mode(FUSED)
    for it in range(3):
        fb.zero_(); mask.fill_(7)
        g.render_frame(fb, mask)
        assert g.frame_status() == 0 and g.frame_mode()[0] == FUSED
        assert same(torch, ref_fb, fb), "single launch, frame %d" % it
        assert torch.equal(ref_mask, mask), "single launch, frame %d: mask" % it
    g.set_frame_mode(AUTO)
    modes = []
    for it in range(70):
        check = it < 4 or it >= 62
        if check:
            fb.zero_(); mask.fill_(7)
        g.render_frame(fb, mask)
        if check:
            assert g.frame_status() == 0
            assert same(torch, ref_fb, fb), "left to choose, frame %d (mode %d)" % (it, g.frame_mode()[0])
            assert torch.equal(ref_mask, mask), "left to choose, frame %d: mask" % it
        modes.append(g.frame_mode()[0])
    assert g.frame_status() == 0
    assert set(modes) == {SPLIT}, "frames of more than 65 536 tiles take three launches by rule"


@pytest.mark.parametrize("size,parts", [(4096, 2), (4096, 4), (8192, 2), (8192, 4)])
def test_sharded_frame_with_the_band_heights_in_use(ra, torch_cuda, size, parts):
    """Rows dealt in parallel.band_height() bands (256 / 128 rows at these sizes, not the 64 of the other tests), every
    part rendered through rtx_render_frame as parallel.shard_frame does (cold + warm frame): the parts assemble to
    exactly the unsharded frame, and every part's mask is the whole mask on its rows and 0 elsewhere."""
    from rendering_amd import assets, parallel
    torch = torch_cuda
    assets.ensure(["bumpy_250k.obj"])
    g = ra.Scene("scenes/cfg2_smooth_250k.scene", size, size)
    band = parallel.band_height(size, parts)
    assert band in (128, 256)
    full, full_mask = stages(torch, g)
    acc = torch.zeros_like(full)
    fb = torch.zeros_like(full); mask = torch.zeros_like(full_mask)
    g.set_frame_mode(AUTO)
    for part in range(parts):
        for it in range(2):
            mask.fill_(5)
            parallel.shard_frame(g, fb, mask, parts, part)
            assert g.frame_status() == 0
        rows = torch.as_tensor(parallel.owned_rows(size, band, parts, part), device="cuda")
        acc.index_copy_(0, rows, fb.index_select(0, rows))
        want = torch.zeros_like(full_mask)
        want.index_copy_(0, rows, full_mask.index_select(0, rows))
        assert torch.equal(want, mask), "mask of part %d of %d" % (part, parts)
    g.set_row_ownership(0, 1, 0, False)
    assert same(torch, full, acc), "%d parts in %d-row bands" % (parts, band)


def test_cfg4_4096_with_ssaa_against_oracle_bands(ra, oracle, torch_cuda):
    """BASELINE cfg4 at 4096^2 WITH the adaptive 4-ray pass (the other cfg4 test stops after pass 1): Sobel mask of the whole
    frame and the re-rendered pixels of row bands against the oracle."""
    from tests.test_gpu_fullsize import check_bands_against_oracle
    from rendering_amd import assets
    assets.ensure()
    g = ra.Scene("scenes/cfg4_textured_1024.scene")
    o = oracle.OracleScene("scenes/cfg4_textured_1024.scene")
    n = check_bands_against_oracle(torch_cuda, g, o, [(1024, 1032), (1700, 1708), (2040, 2048), (2600, 2608)])
    assert n > 1000


def test_frame_kernel_gives_up_and_the_frame_is_rendered_again(ra, torch_cuda):
    """Item queues of ONE entry (knob): the single launch overflows them and gives up; rtx_frame_status then renders the
    frame again through the three launches -- the caller never sees a partial frame (scene.cpp:595-606) -- reports
    3 | 0x100, and the view stays on three launches."""
    torch = torch_cuda
    g = ra.Scene("scenes/cfg2_smooth_4k.scene", 320, 240)
    ref_fb, ref_mask = stages(torch, g)
    assert int(ref_mask.sum()) > 500
    g.set_knob("frame_queue_cap", 1)
    fb = torch.zeros_like(ref_fb); mask = torch.full_like(ref_mask, 9)
    g.set_frame_mode(FUSED)
    g.render_frame(fb, mask)
    assert g.frame_mode()[0] == FUSED
    st = g.frame_status()
    assert st == (3 | 0x100), hex(st)
    assert same(torch, ref_fb, fb) and torch.equal(ref_mask, mask)
    assert g.frame_status() == 0           # nothing pending any more
    g.set_frame_mode(AUTO)
    for it in range(6):                    # the measured choice does not go back to the launch that gave up
        fb.zero_()
        g.render_frame(fb, mask)
        assert g.frame_mode()[0] == SPLIT
    assert g.frame_status() == 0 and same(torch, ref_fb, fb)
    g.set_knob("frame_queue_cap", 0)
    g.set_frame_mode(FUSED)                # forced: the single launch again, with queues of the usual size
    fb.zero_(); mask.fill_(9)
    g.render_frame(fb, mask)
    assert g.frame_status() == 0 and same(torch, ref_fb, fb) and torch.equal(ref_mask, mask)
    g.set_frame_mode(AUTO)


def test_mask_rows_of_other_parts_are_zero_in_either_mode(ra, torch_cuda):
    """rtx_render_frame writes mask rows [row_begin, row_end) completely, 0 for rows owned by another part -- in three
    launches as in one (the three-launch path left them stale); and the last row alone, single launch."""
    from rendering_amd import parallel
    torch = torch_cuda
    W, H = 200, 330
    g = ra.Scene("scenes/cfg2_smooth_4k.scene", W, H)
    _, full_mask = stages(torch, g)
    for mode in (SPLIT, FUSED):
        g.set_frame_mode(mode)
        for parts in (2, 3):
            for part in range(parts):
                fb = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
                mask = torch.full((H, W), 9, dtype=torch.uint8, device="cuda")
                g.set_row_ownership(64, parts, part, True)
                g.render_frame(fb, mask)
                assert g.frame_status() == 0 and g.frame_mode()[0] == mode
                rows = torch.as_tensor(parallel.owned_rows(H, 64, parts, part), device="cuda")
                want = torch.zeros_like(full_mask)
                want.index_copy_(0, rows, full_mask.index_select(0, rows))
                assert torch.equal(want, mask), "mode %d, part %d of %d" % (mode, part, parts)
    g.set_row_ownership(0, 1, 0, False)
    g.set_frame_mode(FUSED)
    fb = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    mask = torch.full((H, W), 9, dtype=torch.uint8, device="cuda")
    g.render_frame(fb, mask, rows=(H - 1, H))
    assert g.frame_status() == 0
    assert not mask[H - 1].any() and bool((mask[:H - 1] == 9).all())
    g.set_frame_mode(AUTO)


def test_pass1_above_32768_rows(ra, oracle, torch_cuda):
    """Frames taller than 32768 rows: tile rows from 4096 on set bit 28 of a plain tile-list entry, which is also the
    marker of a halo strip -- the kernels only decode strips when the list can hold them.  16 x 33000
    pixels of cfg1, whole and in row bands of two parts, against the oracle."""
    from rendering_amd import parallel
    torch = torch_cuda
    W, H = 16, 33000
    g = ra.Scene("scenes/cfg1_simple_shapes.scene", W, H)
    o = oracle.OracleScene("scenes/cfg1_simple_shapes.scene", W, H)
    ref = o.pass1(rows=(32700, 33000))
    fb = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    for it in range(2):      # cold list, then the list ordered by cost
        fb.zero_()
        g.render_pass1(fb)
        torch.cuda.synchronize()
        got = fb.cpu().numpy()
        assert np.array_equal(bits(ref[32700:]), bits(got[32700:])), "frame %d" % it
    assert got[:32700].any()
    whole = got
    acc = np.zeros_like(whole)
    for part in range(2):
        fb.zero_()
        g.set_row_ownership(64, 2, part, True)
        for it in range(2):
            g.render_pass1(fb)
        torch.cuda.synchronize()
        rows = parallel.owned_rows(H, 64, 2, part)
        acc[rows] = fb.cpu().numpy()[rows]
    g.set_row_ownership(0, 1, 0, False)
    assert np.array_equal(bits(whole), bits(acc))
