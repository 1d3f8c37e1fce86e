"""Output side of the hot path: decoded pixels -> the bytes the serving loop JPEG-encodes.

The reference's frame callback (release_server.py:978-991) copies the decoder's float32 pixels [1, T, 3, H, W] to a
pinned host tensor on a download stream (after waiting on the event recorded behind the decode), normalises them on
the CPU (`add_(1.0).mul_(0.5).clamp_(0.0, 1.0)`) and hands every frame to `TF.to_pil_image(...).save(JPEG)` (:972), whose
float path is `mul(255).byte()` in H x W x C order.  `FrameDownloader` is that callback with the arithmetic moved in front
of the copy (`rtv_pixels_to_rgb8`): the device-to-host transfer carries 1 byte per sample instead of 4 (14.4 MB instead of
57.5 MB per 12-frame block), the CPU does no arithmetic, and the result is bit-identical
(`oracle/vae_oracle.frames_to_rgb8`).  JPEG encoding / the WebSocket stay with the caller (control plane, out of scope).
"""
import torch

from . import ops


class FrameDownloader:
    """Use as `GenerationSession(..., frame_callback=downloader)`.  Every call enqueues conversion + async copy into one
    of `slots` pinned buffers on the download stream and returns a ticket; `fetch(ticket)` waits for that copy only and
    returns uint8 [T, H, W, 3] (a view of the pinned buffer, valid until the slot is reused `slots` calls later)."""

    def __init__(self, device="cuda", slots=2):
        if slots < 1:
            raise ValueError("slots must be >= 1")
        self.device = torch.device(device)
        self.stream = torch.cuda.Stream(device=self.device)     # release_server.py:88-90 download_stream
        self.slots = slots
        self._host = [None] * slots
        self._dev = [None] * slots
        self._done = [None] * slots
        self._shape = [None] * slots
        self._frame_ids = [None] * slots
        self._n = 0

    def __call__(self, pixels, frame_ids=(), event=None):
        if pixels.dim() != 5 or pixels.shape[0] != 1 or pixels.shape[2] != 3:
            raise ValueError("expected decoder pixels [1, T, 3, H, W]")
        T, _, H, W = pixels.shape[1:]
        slot = self._n % self.slots
        if self._done[slot] is not None:
            self._done[slot].synchronize()                      # the slot's previous copy must have landed before reuse
        n = T * H * W * 3
        if self._host[slot] is None or self._host[slot].numel() < n:
            self._host[slot] = torch.empty(n, dtype=torch.uint8, pin_memory=True)
            self._dev[slot] = torch.empty(n, dtype=torch.uint8, device=self.device)
        if event is None:
            event = torch.cuda.Event()
            event.record(torch.cuda.current_stream(self.device))
        self.stream.wait_event(event)                           # :981 download_stream.wait_event(event)
        with torch.cuda.stream(self.stream):
            src = pixels[0].float().contiguous()
            src.record_stream(self.stream)
            rgb = ops.pixels_to_rgb8(src, out=self._dev[slot][:n].view(T, H, W, 3))
            self._host[slot][:n].copy_(rgb.view(-1), non_blocking=True)
            done = torch.cuda.Event()
            done.record(self.stream)
        self._done[slot], self._shape[slot], self._frame_ids[slot] = done, (T, H, W, 3), list(frame_ids)
        self._n += 1
        return self._n - 1

    def fetch(self, ticket):
        if not (self._n - self.slots <= ticket < self._n) or ticket < 0:
            raise KeyError(f"ticket {ticket} is no longer (or not yet) held; {self.slots} slots")
        slot = ticket % self.slots
        self._done[slot].synchronize()
        T, H, W, C = self._shape[slot]
        return self._host[slot][:T * H * W * C].view(T, H, W, C)

    def frame_ids(self, ticket):
        return self._frame_ids[ticket % self.slots]
