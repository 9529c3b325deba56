"""N > 1 path on CPU: world_size-2 gloo run of the row-band sharding + gather (rendering_amd/parallel.py).
The renderer is replaced by the CPU oracle here (tests may use it); on the GPU box the same functions move
device tensors over RCCL."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
from rendering_amd import parallel
from oracle import oracle as O
dist.init_process_group('gloo')
rank, world = dist.get_rank(), dist.get_world_size()
W, H, band = 96, 88, 16           # H is not a multiple of band*world: exercises the ragged last band
o = O.OracleScene('scenes/cfg2_smooth_4k.scene', W, H)
full = o.pass1()
mine = parallel.owned_rows(H, band, world, rank)
fb = np.zeros_like(full)
# what a rank renders: its bands plus the 1-row halo
need = sorted(set(mine) | {y - 1 for y in mine if y > 0} | {y + 1 for y in mine if y + 1 < H})
for y in need:
    fb[y] = o.pass1(rows=(y, y + 1))[y]
mask_full = o.sobel(full)
mask = o.sobel(fb)
assert np.array_equal(mask[mine], mask_full[mine]), 'halo rows make the Sobel mask of owned rows exact'
own = np.zeros((H, W), np.uint8); own[mine] = mask[mine]
fb2 = o.ssaa(fb, own)
keep = np.zeros_like(fb2); keep[mine] = fb2[mine]
t = torch.from_numpy(keep.copy())
parallel.gather_frame(t, world, rank, band=band)
ref = o.ssaa(full, mask_full)
if rank == 0:
    got = t.numpy()
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), 'gathered frame != single-process frame'
    # the quantised, bottom-up image travels the same way (what bench.py gathers at N > 1)
q = np.zeros((H, W, 3), np.uint8)
qk = np.clip(keep, 0, 1)
q[H - 1 - mine] = (qk[mine][:, :, ::-1] * 255).astype(np.uint8)
tq = torch.from_numpy(q)
parallel.gather_frame(tq, world, rank, band=band, bottom_up=True)
if rank == 0:
    want = (np.clip(ref, 0, 1)[::-1, :, ::-1] * 255).astype(np.uint8)
    assert np.array_equal(tq.numpy(), want), 'gathered BGR8 image != single-process image'
    print('GATHER_OK')
dist.destroy_process_group()
"""


def test_two_rank_gloo_shard_and_gather(tmp_path):
    script = tmp_path / "child.py"
    script.write_text(CHILD % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29611", str(script)],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "GATHER_OK" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]


def test_owned_rows_partition():
    from rendering_amd import parallel
    for H, band, n in ((4096, 64, 8), (1080, 64, 4), (88, 16, 2), (10, 64, 3)):
        allrows = np.concatenate([parallel.owned_rows(H, band, n, r) for r in range(n)])
        assert sorted(allrows.tolist()) == list(range(H))


def test_gather_plan_matches_band_ranges(ra):
    """rtx_gather_plan (the transfer list of rtx_gather, C ABI) against the Python statement of the same banding."""
    from rendering_amd import parallel
    for H, band, parts, rb, bu in ((4096, 64, 8, 4096 * 3, True), (203, 64, 2, 316 * 12, False), (1080, 64, 3, 1920 * 3, True), (64, 64, 4, 12, False)):
        plan = ra.gather_plan(H, band, parts, rb, bu)
        want = []
        for y0 in range(0, H, band):
            y1 = min(y0 + band, H)
            want.append(((y0 // band) % parts, (H - y1 if bu else y0) * rb, (y1 - y0) * rb))
        assert plan == want
        for part in range(parts):
            mine = [(o, n) for r, o, n in plan if r == part]
            assert mine == [((H - y1 if bu else y0) * rb, (y1 - y0) * rb) for y0, y1 in parallel.band_ranges(H, band, parts, part)]
        # the slabs tile the image exactly once
        cover = sorted((o, n) for _, o, n in plan)
        assert cover[0][0] == 0 and all(cover[i][0] + cover[i][1] == cover[i + 1][0] for i in range(len(cover) - 1)) and cover[-1][0] + cover[-1][1] == H * rb
