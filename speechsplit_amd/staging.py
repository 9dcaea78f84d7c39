"""Double-buffered host->device staging of the training batches (north_star: "the next
make_spect_f0 mel/F0 batch staged via pinned hipMemcpyAsync").  While the engine runs step i on the compute stream, the
collated batch i+1 is copied from pinned host memory on a second HIP stream; the compute stream only waits on the copy's
event.  One batch is 64 x 128 x (80 + 1) floats + embeddings = 2.7 MB: 43 us at PCIe Gen5, fully hidden behind a 15 ms step.
"""
import torch


class DevicePrefetcher:
    def __init__(self, loader, device):
        self.loader = loader
        self.device = torch.device(device)
        self.stream = torch.cuda.Stream(self.device)
        self.it = iter(loader)
        self.ready = None
        self._stage()

    def _stage(self):
        try:
            batch = next(self.it)
        except StopIteration:            # solver.py:141-145 restarts the iterator on exhaustion
            self.it = iter(self.loader)
            batch = next(self.it)
        with torch.cuda.stream(self.stream):
            # host batches: pinned + non-blocking copy; batches assembled on the GPU (DeviceBatcher) are already there and
            # their assembly kernel ran on this stream
            self.ready = tuple(t if t.is_cuda else (t if t.is_pinned() else t.pin_memory()).to(self.device, non_blocking=True)
                               for t in batch)

    def __iter__(self):
        return self

    def __next__(self):
        cur = torch.cuda.current_stream(self.device)
        cur.wait_stream(self.stream)
        batch = self.ready
        for t in batch:
            t.record_stream(cur)
        self._stage()
        return batch


class DrawStager:
    """Per-step upload of the InterpLnr draws (reference model.py:392-393, 399-402: rand(B*7) + 0.5, randint per call) without
    stalling the host: the draws of a step are packed into ONE pinned buffer of a small ring ([ncalls][n] f32 scales followed by
    [ncalls][n] i32 segment lengths, 7 KB at batch 64) and copied with one non-blocking hipMemcpyAsync on the compute stream,
    i.e. ordered in front of the step that reads them and behind the previous step's kernels (which read the same device
    buffer).  A slot is reused only after the copy that read it has completed (an event per slot; by then it is `depth` steps
    old, so the wait never blocks in practice).  The pageable `.to(device)` this replaces made every step start with a
    blocking copy, i.e. an implicit host-device synchronisation."""

    def __init__(self, device, ncalls, n, depth=8):
        self.device = torch.device(device)
        self.ncalls, self.n, self.depth = ncalls, n, depth
        self.host = [torch.empty(2 * ncalls * n, dtype=torch.float32).pin_memory() for _ in range(depth)]
        self.events = [None] * depth
        self.dev = torch.empty(2 * ncalls * n, dtype=torch.float32, device=self.device)
        self.i = 0

    def stage(self, scales, len_seg):
        """scales f32 [ncalls, n], len_seg int [ncalls, n] on the host -> (scales_dev f32, len_seg_dev i32) views."""
        k = self.i % self.depth
        self.i += 1
        if self.events[k] is not None:
            self.events[k].synchronize()
        h = self.host[k]
        m = self.ncalls * self.n
        h[:m].view(self.ncalls, self.n).copy_(torch.as_tensor(scales, dtype=torch.float32))
        h[m:].view(torch.int32).view(self.ncalls, self.n).copy_(torch.as_tensor(len_seg).to(torch.int32))
        self.dev.copy_(h, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self.events[k] = ev
        return self.dev[:m].view(self.ncalls, self.n), self.dev[m:].view(torch.int32).view(self.ncalls, self.n)
