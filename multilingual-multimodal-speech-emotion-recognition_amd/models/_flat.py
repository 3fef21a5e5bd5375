"""Flat parameter / gradient storage for the trainable head modules.

Each module keeps its grad-bearing parameters as views into ONE contiguous fp32 buffer, with a
matching flat gradient buffer.  That gives (a) one RCCL all-reduce per module bucket instead of one
per tensor, (b) one fused AdamW launch per optimizer group, and (c) gradient kernels that write
straight into the bucket (no autograd-side copies).  `state_dict()` / `load_state_dict()` are
unaffected because the nn.Parameters stay in place (their `.data` is re-pointed at the views).
"""
import weakref

import torch

ALIGN = 64  # floats (256 B): keeps every view aligned for float4 kernels
BUCKETS = weakref.WeakSet()   # every live bucket, so the optimizer / DP reducer can find a parameter's home


def find_bucket(p):
    for b in BUCKETS:
        for i, q in enumerate(b.params):
            if q is p:
                return b, i
    return None, -1


class FlatParams:
    def __init__(self, params):
        self.params = [p for p in params]
        self.flat = None
        self.gflat = None
        self.offsets = []
        off = 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        self.total = off
        self.grad_ready_hook = None     # called with the bucket once its gradients are complete (DP overlap)
        BUCKETS.add(self)

    def _views(self, buf):
        return [buf[o:o + p.numel()].view(p.shape) for o, p in zip(self.offsets, self.params)]

    def ensure(self):
        """(Re)flatten when the parameters are not (or no longer) views of the flat buffer."""
        if not self.params:
            return self
        dev = self.params[0].device
        ok = self.flat is not None and self.flat.device == dev
        if ok:
            base = self.flat.data_ptr()
            ok = all(p.data_ptr() == base + 4 * o for p, o in zip(self.params, self.offsets))
        if not ok:
            # Stream safety.  Re-pointing `p.data` drops the last reference to the parameter's old storage, and the
            # caching allocator hands a freed block back to its ALLOCATION stream at once.  When this runs on another
            # stream (the text-side modules of SERSystem flatten lazily on its side stream) the copies below are still
            # queued there while the main stream's next allocation — the sibling module's flat buffer has exactly the
            # size of the merged freed blocks — may reuse and overwrite the block: the parameters then silently become
            # zeros or the sibling's values (the rare first-loss mismatch of round 2's two-rank rehearsal, DESIGN.md
            # section 6).  `record_stream` makes the allocator wait for this stream before reusing the old blocks, and
            # the old flat buffers of a re-flatten are retired the same way.
            cuda = dev.type == "cuda"
            cur = torch.cuda.current_stream(dev) if cuda else None
            flat = torch.zeros(self.total, dtype=torch.float32, device=dev)
            for v, p in zip(self._views(flat), self.params):
                old = p.data
                v.copy_(old)
                if cuda and old.is_cuda:
                    old.record_stream(cur)
                p.data = v
                del old
            for old in (self.flat, self.gflat):
                if cuda and old is not None and old.is_cuda:
                    old.record_stream(cur)
            self.flat = flat
            self.gflat = torch.zeros(self.total, dtype=torch.float32, device=dev)
            self.gviews = self._views(self.gflat)
            for p in self.params:
                p.grad = None
            # consumers on OTHER streams (the optimizer, a captured graph) must not run ahead of the copies above: a
            # (re)flatten is a rare set-up event, so outside graph capture simply drain the device once
            if cuda and not torch.cuda.is_current_stream_capturing():
                torch.cuda.synchronize(dev)
        return self

    def gview(self, p):
        return self.gviews[self._index(p)]

    def _index(self, p):
        for i, q in enumerate(self.params):
            if q is p:
                return i
        raise KeyError("parameter is not part of this flat bucket")

    def accumulating(self):
        """True when the bucket already holds this step's gradients (a second backward before zero_grad)."""
        return all(p.grad is not None and p.grad.data_ptr() == g.data_ptr() for p, g in zip(self.params, self.gviews))

    def publish(self):
        for p, g in zip(self.params, self.gviews):
            if p.requires_grad:
                p.grad = g
        if self.grad_ready_hook is not None:
            self.grad_ready_hook(self)

    def padded_numel(self, i):
        nxt = self.offsets[i + 1] if i + 1 < len(self.offsets) else self.total
        return nxt - self.offsets[i]
