"""Pixel sharding of one frame over the GPUs of a node.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).  Rows are dealt to ranks in
round-robin bands of BAND rows (the mesh sits centre-frame, so contiguous blocks would be unbalanced).
Per frame and rank:
    pass 1 on the owned bands + a 1-row halo (recomputed, so the Sobel mask needs no exchange)
    Sobel mask + adaptive 4-ray pass on the owned rows
    ONE collective: gather of the owned rows to rank 0 (the only real exchange step of the path).
The scene (<= ~60 MB) is replicated on every GPU.
"""
import numpy as np

BAND = 64


def owned_rows(height, band, n_parts, part):
    """Row indices y with (y // band) % n_parts == part (the ownership rule of rtx_set_row_ownership)."""
    y = np.arange(height)
    return y[(y // band) % n_parts == part]


def shard_frame(scene, fb, mask, n_parts, part, band=BAND, ssaa=True, stream=None):
    """Renders this rank's rows of one frame into the device tensors fb (H,W,3 f32) / mask (H,W u8)."""
    fb.zero_()
    scene.set_row_ownership(band if n_parts > 1 else 0, n_parts, part, halo=True)
    scene.render_pass1(fb, stream=stream)
    if ssaa:
        scene.sobel(fb, mask, stream=stream)
        scene.render_ssaa(mask, fb, stream=stream)


def gather_frame(fb, n_parts, part, band=BAND, dst=0, group=None):
    """Collects every rank's owned rows into rank `dst`'s fb (in place).  Works for device tensors over RCCL
    and for CPU tensors over gloo (the world_size-2 CPU test)."""
    import torch
    import torch.distributed as dist
    if n_parts == 1:
        return fb
    H = fb.shape[0]
    counts = [len(owned_rows(H, band, n_parts, r)) for r in range(n_parts)]
    mx = max(counts)
    mine = torch.as_tensor(owned_rows(H, band, n_parts, part), device=fb.device)
    send = torch.zeros((mx,) + tuple(fb.shape[1:]), dtype=fb.dtype, device=fb.device)
    send[: len(mine)] = fb.index_select(0, mine)
    if part == dst:
        recv = [torch.empty_like(send) for _ in range(n_parts)]
        dist.gather(send, recv, dst=dst, group=group)
        for r in range(n_parts):
            if r == dst:
                continue
            rows = torch.as_tensor(owned_rows(H, band, n_parts, r), device=fb.device)
            fb.index_copy_(0, rows, recv[r][: len(rows)])
    else:
        dist.gather(send, None, dst=dst, group=group)
    return fb
