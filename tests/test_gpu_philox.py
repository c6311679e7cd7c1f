"""In-kernel noise == torch's (SURVEY.md section 7 "reproduce torch's Philox offsets").

The reference's sampler consumes a device torch.Generator through torch.multinomial (== argmax(p / exponential_)) and
torch.rand (JL:118, 237, 260).  K2 / K4 generate those values in place (csrc/sjd_philox.h); here the same device functions write whole
tensors (sjd_philox_fill) and are compared BIT FOR BIT with `torch.empty(...).exponential_(generator=g)` / `torch.rand(..., generator=g)`
drawn on the GPU -- for the sizes the decode uses (1 / 5 / 16 / 32 rows of V = 16384 / 65536 / 184622), sizes that are no multiple of a
block, and offsets that a long decode reaches -- and the generator's offset must advance by sjd_philox_offset_increment.
"""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu

SIZES = [1, 7, 255, 256, 257, 1000, 16384, 65536, 5 * 65536, 16 * 65536, 184622, 3 * 184622, 32 * 184622, 16 * 16384, 2048 * 256 * 4 + 1]


def _max_blocks(dev):
    p = torch.cuda.get_device_properties(dev)
    return p.multi_processor_count * (p.max_threads_per_multi_processor // 256)


@pytest.mark.parametrize("kind", ["rand", "exponential"])
def test_philox_fill_equals_torch(kind):
    import sjd_amd._lib as L
    from sjd_amd.ops import philox_max_blocks
    lib = L.load()
    dev = torch.device("cuda:0")
    mb = philox_max_blocks(dev)
    assert mb == _max_blocks(dev)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for seed in (0, 42, 1234, 2 ** 40 + 17):
        g = torch.Generator(dev).manual_seed(seed)
        assert g.initial_seed() == seed and g.get_offset() == 0
        for numel in SIZES:
            for extra in (0, 4 * 1000003, 2 ** 33 + 8):
                g.set_offset(g.get_offset() + extra)
                off0 = g.get_offset()
                if kind == "rand":
                    want = torch.rand(numel, generator=g, device=dev)
                else:
                    want = torch.empty(numel, device=dev).exponential_(generator=g)
                assert g.get_offset() - off0 == lib.sjd_philox_offset_increment(numel, mb), (numel, g.get_offset() - off0)
                got = torch.empty(numel, device=dev)
                L.check(lib.sjd_philox_fill(got.data_ptr(), numel, seed, off0, mb, 0 if kind == "rand" else 1, stream), "sjd_philox_fill")
                torch.cuda.synchronize()
                same = torch.equal(got.view(torch.int32), want.view(torch.int32))
                if not same:
                    bad = (got.view(torch.int32) != want.view(torch.int32)).nonzero()[:4, 0].tolist()
                    raise AssertionError(f"{kind} seed {seed} numel {numel} offset {off0}: first differences at {bad}: "
                                         f"{[(float(got[i]), float(want[i])) for i in bad]}")


def test_multinomial_is_argmax_over_in_kernel_exponential():
    """the equivalence the kernels rest on, now with the in-kernel noise: torch.multinomial(p, 1, generator=g) ==
    argmax(p / E) with E = the tensor sjd_philox_fill writes for the same generator state, and the generator ends in the same state"""
    import sjd_amd._lib as L
    from sjd_amd.ops import philox_max_blocks
    lib = L.load()
    dev = torch.device("cuda:0")
    mb = philox_max_blocks(dev)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for seed, rows, V in ((3, 16, 65536), (4, 1, 16384), (5, 32, 184622)):
        g = torch.Generator(dev).manual_seed(seed)
        p = torch.softmax(torch.randn(rows, V, device=dev) * 3, dim=-1)
        g.set_offset(4 * 777)
        off0 = g.get_offset()
        want = torch.multinomial(p, 1, generator=g)[:, 0]
        E = torch.empty(rows, V, device=dev)
        L.check(lib.sjd_philox_fill(E.data_ptr(), rows * V, seed, off0, mb, 1, stream), "sjd_philox_fill")
        assert torch.equal((p / E).argmax(-1), want)
        assert g.get_offset() - off0 == lib.sjd_philox_offset_increment(rows * V, mb)
