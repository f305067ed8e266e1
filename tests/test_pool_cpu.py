"""engine.BufferPool (the opt-in shared activation pool of the two passes) on the CPU: allocation order, alignment, aliasing after a
rewind, first fit into the room a larger request left, growth.  The GPU side (pad rows re-zeroed, same gradients as private buffers,
misuse errors) is tests/test_train_gpu.py::test_shared_activation_pool_is_the_same_training."""
import torch

from avsiam_amd.engine import BufferPool


def test_pool_allocates_aligned_views_and_aliases_after_rewind():
    pool = BufferPool(torch.device("cpu"))
    pool.CHUNK = 1 << 16                                   # small chunks so that the test crosses chunk boundaries
    a = pool.alloc((100, 7), torch.float32)
    b = pool.alloc((33, 5), torch.bfloat16)
    c = pool.alloc((3000,), torch.uint8)
    def offset(t):                                         # byte offset inside the chunk that holds it (the chunk base is the allocator's: 512-byte aligned on the GPU)
        for ch in pool.chunks:
            if ch.data_ptr() <= t.data_ptr() < ch.data_ptr() + ch.numel():
                return t.data_ptr() - ch.data_ptr()
        raise AssertionError("not in the pool")
    for t in (a, b, c):
        assert offset(t) % 256 == 0 and t.is_contiguous() and float(t.float().abs().sum()) == 0
    assert a.shape == (100, 7) and b.dtype == torch.bfloat16
    # disjoint
    spans = sorted((t.data_ptr(), t.data_ptr() + t.numel() * t.element_size()) for t in (a, b, c))
    assert all(spans[i][1] <= spans[i + 1][0] for i in range(2))
    a.fill_(1.0); b.fill_(2.0); c.fill_(3)
    # the next pass starts at the first byte again: same addresses for the same request sequence, whatever it left there
    pool.rewind()
    a2 = pool.alloc((100, 7), torch.float32)
    assert a2.data_ptr() == a.data_ptr() and float(a2.sum()) == 700.0            # NOT zeroed: the other pass's data (Stack re-zeroes the pad rows)
    # a different sequence aliases too, and a request larger than any chunk opens its own
    pool.rewind()
    big = pool.alloc((1 << 18,), torch.float32)
    assert big.numel() == 1 << 18 and pool.nbytes() >= 4 << 18
    small = pool.alloc((16,), torch.float32)
    assert small.data_ptr() == a.data_ptr()                                       # first fit: the first chunk is still empty in this pass
    n = pool.nbytes()
    pool.rewind()
    pool.alloc((1 << 18,), torch.float32)
    assert pool.nbytes() == n                                                     # nothing new is allocated for a sequence that fitted before
