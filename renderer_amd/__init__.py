"""renderer_amd — MI355X-native deferred-lighting pass for dobrakmato/renderer's
screen-space hot path (lighting subpass + tonemap subpass).

The product is ``libmil.so`` (hand-written HIP for gfx950 behind the C ABI in
``include/mil.h``); this package is the host-side mirror used by the tests and
the benchmark.  Nothing here falls back to a CPU path: calls raise if the HIP
library is missing.
"""
from . import ubo  # noqa: F401

__all__ = ['ubo']
