"""Pixel sharding of one frame over the GPUs of a node.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).  Rows are dealt to ranks in
round-robin bands of band_height() rows (the mesh sits centre-frame, so contiguous blocks would be unbalanced).
Per frame and rank:
    pass 1 on the owned bands + a 1-row halo (recomputed, so the Sobel mask needs no exchange)
    Sobel mask + adaptive 4-ray pass on the owned rows
    quantise to BGR8 (what saveImage writes, util.cpp:46-58; 4x fewer bytes than the fp32 framebuffer)
    ONE exchange: every owned band is sent to rank 0 as a point-to-point message, placed directly -- rtx_gather of the
    C ABI (RCCL, include/rtx.h); make_comm() bootstraps its communicator over an existing torch.distributed group.
In a sequence of frames the exchange of frame k runs on a second HIP stream while frame k + 1 is rendered (FrameGather):
a part's stages last as long as their slowest wave, not as long as their work (tools/shard_stages.py: one eighth of the
headline frame's SSAA takes 0.37 ms, the whole 0.51), so cutting a frame into band groups to send the first while the second
renders would pay those tails once per group -- frames, not bands, are what is pipelined.
The scene (<= ~60 MB) is replicated on every GPU.  gather_frame() is the same exchange over torch.distributed
point-to-point operations; it exists for the CPU (gloo) tests and for boxes where the ranks have to share a device.
"""
import numpy as np

BAND = 64      # the smallest band; band_height() is what the frame is dealt in


def band_height(height, n_parts):
    """Rows per band for a frame of `height` rows over n_parts devices: about eight bands per device (the mesh sits
    centre-frame, a device needs rows from everywhere), between 64 and 256 rows -- every band costs two halo rows, and
    measured on one device's share (tools/shard_time.py) 256-row bands beat 64-row ones by 8 points of efficiency at
    N = 2, while at N = 8 a 4096-row frame needs the small ones for balance.  Scene::render() of the C++ host uses the
    same rule (host/src/scene.cpp)."""
    return int(min(256, max(BAND, (height // (8 * max(n_parts, 1))) // 64 * 64)))


def owned_rows(height, band, n_parts, part):
    """Row indices y with (y // band) % n_parts == part (the ownership rule of rtx_set_row_ownership)."""
    y = np.arange(height)
    return y[(y // band) % n_parts == part]


def shard_frame(scene, fb, mask, n_parts, part, band=None, ssaa=True, stream=None, clear=True):
    """Renders this rank's rows of one frame into the device tensors fb (H,W,3 f32) / mask (H,W u8).
    clear: zero fb first, like the reference's `new Vec3f[H*W]()` (scene.cpp:599, outside its "Render scene" timer) -- the
    last row / column and the rows of other ranks are never written; a caller that re-renders into a buffer that already
    is such a frame (bench.py) passes False."""
    if clear:
        fb.zero_()
    if band is None:
        band = band_height(fb.shape[0], n_parts)
    scene.set_row_ownership(band if n_parts > 1 else 0, n_parts, part, halo=True)
    if ssaa:
        scene.render_frame(fb, mask, stream=stream)      # pass 1 + Sobel + SSAA (rtx_render_frame: one launch or three)
    else:
        scene.render_pass1(fb, stream=stream)


def band_ranges(height, band, n_parts, part):
    """[(y0, y1), ...]: the contiguous row ranges (bands) rank `part` owns."""
    return [(y0, min(y0 + band, height)) for b, y0 in enumerate(range(0, height, band)) if b % n_parts == part]


def gather_frame(img, n_parts, part, band=None, dst=0, group=None, bottom_up=False):
    """Collects every rank's owned rows into rank `dst`'s img (in place), with no staging copies: a band is a
    contiguous slab of img, so every band travels as one point-to-point message straight from the owner's buffer into
    its final place in dst's buffer (xGMI links are point-to-point: one grouped batch of sends / receives, no ring).
    img: (H, ...) tensor -- the fp32 framebuffer, or the quantised BGR8 image with bottom_up=True (rtx_quantize_bgr8
    stores image row y at H-1-y, util.cpp:50).  Device tensors go over RCCL, CPU tensors over gloo (CPU test)."""
    import torch.distributed as dist
    if n_parts == 1:
        return img
    H = img.shape[0]
    flat = img.view(H, -1)
    if band is None:
        band = band_height(H, n_parts)

    def slab(y0, y1):
        return flat[H - y1:H - y0] if bottom_up else flat[y0:y1]

    ops = []
    if part == dst:
        for r in range(n_parts):
            if r != dst:
                ops += [dist.P2POp(dist.irecv, slab(y0, y1), r, group) for y0, y1 in band_ranges(H, band, n_parts, r)]
    else:
        ops = [dist.P2POp(dist.isend, slab(y0, y1), dst, group) for y0, y1 in band_ranges(H, band, n_parts, part)]
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return img


class FrameGather:
    """The exchange of one frame overlapped with the rendering of the next.  submit(fb), called after a frame's kernels
    were queued on the current stream, queues there the quantiser (this rank's rows of fb -> img, BGR8 bottom-up) and on a
    second stream rtx_gather of img; the next frame's kernels do not wait for it.  img is reused: the quantiser of frame
    k + 1 waits (on the device, an event) until the exchange of frame k has left / filled img -- long after it ended, as
    a frame renders several times longer than it travels.  wait() makes the current stream wait for the last exchange:
    rank `root`'s img then holds the last frame submitted.  A consumer that needs every frame calls wait() per frame (and
    gives up the overlap) or hands in a different img per frame.

    A frame rendered through rtx_render_frame must be CONFIRMED before its image is used: submit() quantises and sends what the
    frame's launches wrote, before the host has seen rtx_frame_status.  If the single launch gave up, scene.frame_status() renders
    the frame again -- into fb only; the caller must then submit(fb) again (bench.py checks the status after its timed region and
    voids the run).  Callers that cannot re-submit force three launches first: scene.set_frame_mode(0)."""

    def __init__(self, scene, comm, img, root=0):
        import torch
        self.scene, self.comm, self.img, self.root = scene, comm, img, root
        self.side = torch.cuda.Stream()
        self.done = None

    def submit(self, fb):
        import torch
        cur = torch.cuda.current_stream()
        if self.done is not None:
            cur.wait_event(self.done)
        self.scene.quantize(fb, self.img)
        ready = torch.cuda.Event()
        ready.record(cur)
        self.side.wait_event(ready)
        self.comm.gather(self.scene, self.img, bottom_up=True, root=self.root, stream=self.side)
        self.done = torch.cuda.Event()
        self.done.record(self.side)

    def wait(self):
        import torch
        if self.done is not None:
            torch.cuda.current_stream().wait_event(self.done)


def make_comm(n_parts, part, device, group=None):
    """rendering_amd.Comm (rtx_comm_create) for the ranks of a torch.distributed group: rank 0's RCCL id is broadcast
    as a byte tensor over that group (any backend)."""
    import torch
    import torch.distributed as dist
    import rendering_amd as RA

    def exchange(raw):
        dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
        t = torch.tensor(list(raw), dtype=torch.uint8, device=dev) if raw is not None else torch.zeros(128, dtype=torch.uint8, device=dev)
        dist.broadcast(t, src=0, group=group)
        return bytes(t.cpu().tolist())

    return RA.Comm(n_parts, part, device, exchange)
