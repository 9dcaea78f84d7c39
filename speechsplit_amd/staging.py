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
            self.ready = tuple((t if t.is_pinned() else t.pin_memory()).to(self.device, non_blocking=True) for t in batch)

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
