"""Stack-discipline device arena for the activations of one forward (caller-owned memory: the C ABI never allocates).

Blocks are plain torch uint8 tensors; `alloc` bump-allocates 256-B aligned views, `mark`/`release` free everything
allocated after the mark.  After one warm-up forward the block list is stable, so a captured HIP graph replays
against fixed addresses."""
import torch

_ALIGN = 256
_ESIZE = {torch.bfloat16: 2, torch.float16: 2, torch.float32: 4, torch.uint8: 1, torch.int32: 4}


class Workspace:
    def __init__(self, device, chunk_bytes=2 << 30):
        self.device = torch.device(device)
        self.chunk_bytes = chunk_bytes
        self.blocks = []          # list of uint8 tensors
        self.cur = 0              # index of the active block
        self.off = 0              # bump offset inside the active block
        self.peak = 0
        self.el = torch.bfloat16  # element type of the forward using this arena (default dtype of alloc)
        self.split = False        # residual trunk stored as hi + lo planes (trunk_dtype "fp16x2")

    def _block_for(self, nbytes):
        while True:
            if self.cur < len(self.blocks):
                blk = self.blocks[self.cur]
                if self.off + nbytes <= blk.numel():
                    return blk
                self.cur += 1
                self.off = 0
                continue
            size = max(self.chunk_bytes, nbytes)
            self.blocks.append(torch.empty(size, dtype=torch.uint8, device=self.device))

    def alloc(self, shape, dtype=None):
        dtype = self.el if dtype is None else dtype
        n = 1
        for s in shape:
            n *= int(s)
        nbytes = n * _ESIZE[dtype]
        nbytes_al = (nbytes + _ALIGN - 1) // _ALIGN * _ALIGN
        blk = self._block_for(nbytes_al)
        t = blk[self.off:self.off + nbytes].view(dtype).view(*shape)
        self.off += nbytes_al
        used = sum(b.numel() for b in self.blocks[:self.cur]) + self.off
        if used > self.peak:
            self.peak = used
        return t

    def trunk(self, shape):
        """A RESIDUAL-TRUNK tensor (block inputs / outputs, everything a branch result is added back into): one plane, or --
        split mode -- the hi plane carrying its lo plane as the attribute `.lo` (value = hi + lo, the lo plane one e5m2 byte
        per element; GEMM A operands read the hi plane in place, residual operands and norm inputs read both: csrc/plan.hip
        `Trk`)."""
        t = self.alloc(shape)
        t.lo = self.alloc(shape, torch.uint8) if self.split else None
        return t

    def mark(self):
        return (self.cur, self.off)

    def release(self, mark):
        self.cur, self.off = mark

    def reset(self):
        self.cur, self.off = 0, 0

    def capacity(self):
        return sum(b.numel() for b in self.blocks)
