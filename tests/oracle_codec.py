"""TEST-ONLY stand-in for garage_amd.ReedSolomon on boxes without a GPU: same
method signatures, arithmetic done by the CPU oracle.  It lets the CPU suite
exercise host-side logic (striping, all-gather plumbing, BlockManager fan-out)
over `gloo`.  The product never imports this; it lives under tests/."""
from __future__ import annotations

import numpy as np

from oracle import rs_oracle as O


class OracleCodec:
    def __init__(self, k: int, m: int):
        self.k, self.m, self.n = k, m, k + m

    # device-style API on CPU tensors --------------------------------------
    def reconstruct_scattered_dev(self, buf, nblocks, block_stride, shard_off, S, present, data_only=False,
                                  byte_range=None):
        flat = buf.numpy()  # shares memory with the CPU tensor
        off, ln = (0, S) if byte_range is None else byte_range
        for b in range(nblocks):
            base = b * block_stride
            st = np.stack([flat[base + shard_off[j] + off: base + shard_off[j] + off + ln] for j in range(self.n)])
            fixed = O.reconstruct(self.k, self.m, st, present, data_only=data_only)
            for j in range(self.n):
                if not present[j] and not (data_only and j >= self.k):
                    flat[base + shard_off[j] + off: base + shard_off[j] + off + ln] = fixed[j]
        return buf

    # host API ----------------------------------------------------------------
    def encode_blocks(self, blocks, S=None):
        if S is None:
            S = max(O.shard_len(self.k, len(b)) for b in blocks)
        return [O.encode(self.k, self.m, O.split_block(self.k, b, S)) for b in blocks]

    def reconstruct(self, shards, data_only=False):
        out = []
        for row in shards:
            S = next(len(s) for s in row if s is not None)
            st = np.stack([np.zeros(S, dtype=np.uint8) if s is None else np.asarray(s, dtype=np.uint8) for s in row])
            present = [s is not None for s in row]
            fixed = O.reconstruct(self.k, self.m, st, present, data_only=data_only)
            out.append([None if (row[j] is None and data_only and j >= self.k) else fixed[j] for j in range(self.n)])
        return out

    def reconstruct_data(self, shards):
        return self.reconstruct(shards, data_only=True)

    def verify(self, stripes):
        return np.array([O.verify(self.k, self.m, st) for st in np.asarray(stripes)], dtype=bool)
