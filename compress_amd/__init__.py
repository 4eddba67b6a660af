"""compress_amd -- MI355X-native XFLATE/DEFLATE read path (drop-in for dsnet/compress xflate/flate readers).

  compress_amd.xflate   NewReader / Reader.Read / Seek / Close / Reset  (xflate/reader.go)
                        NewWriter / Writer (stream generator; xflate/writer.go)
  compress_amd.flate    NewReader / Reader.Read / Close                  (flate/reader.go)
  compress_amd.errors   the reference's error classes                    (internal/errors/errors.go)
  compress_amd.capi     ctypes binding of include/xflate_amd.h (libxflate_amd.so)

The decode work is done by hand-written HIP kernels behind the C ABI; there is
no CPU fallback: without the built library or without a GPU the readers raise.
"""
from . import errors  # noqa: F401

__all__ = ["errors", "capi", "xflate", "flate"]
