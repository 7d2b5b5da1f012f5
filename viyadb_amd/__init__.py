"""viyadb_amd — MI355X-native scan / filter / aggregate executor for ViyaDB's
``query::AggregateQuery`` path (reference: src/query/runner.cc:45-64 ->
src/codegen/query/agg_query.cc:26-75).

Layout (only what the path needs):
  csrc/        hand-written gfx950 HIP kernels + the C-ABI (include/viya_hip.h)
  capi.py      ctypes mirror of the C-ABI
  executor.py  thin Python handle (device table mirror + aggregate call)
  synth.py     the synthetic workloads C1..C5 of SURVEY.md §8(d) (data definitions only)
  build.py     in-tree hipcc build of libviya_hip.so (+ the C++ host shim)
There is no CPU fallback anywhere in this package.
"""
__version__ = "0.1.0"
