"""numpy restatement of the library's counter-based dense generator (bof_gen_dense mode 'u',
blas-on-flash_amd/csrc/gen_kernels.hip): x[g] = u24(splitmix64(seed ^ g * 0xD1342543DE82EF95)) * 2^-23 - 1,
uniform in [-1, 1) with 24 random mantissa bits -- every step is exact in fp32, so host and device agree
bit for bit.  Used to regenerate the inputs of tests/golden/mkl_golden_big.npz instead of storing them."""
import numpy as np


def dense_u(first, count, seed):
    with np.errstate(over="ignore"):
        g = np.arange(first, first + count, dtype=np.uint64)
        x = np.uint64(seed) ^ (g * np.uint64(0xD1342543DE82EF95))
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    r = (x >> np.uint64(40)).astype(np.uint32)
    return r.astype(np.float32) * np.float32(1.0 / 8388608.0) - np.float32(1.0)
