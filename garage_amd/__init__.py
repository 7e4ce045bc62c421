"""garage_amd -- MI355X-native Reed-Solomon erasure-coding engine for Garage's
object-block write/read path.  The product is
`libgarage_ec.so` (HIP kernels + C ABI, include/garage_ec.h); this package is
the thin Python host layer used by tests and bench.py."""
from ._lib import GecError, LIB_PATH  # noqa: F401
from .codec import ReedSolomon, build_decode_matrix, build_matrix, set_kernel_variant, shard_len, shardsum  # noqa: F401
from .group import Group  # noqa: F401
