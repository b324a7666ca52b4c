"""llm.f90 decode hot path, MI355X-native.

Only what the path needs lives here:
  csrc/   hand-written gfx950 HIP kernels + the C-ABI shim (libllmk.so, include/llmk.h)
  host/   Fortran host: weight_module types, GGUF loader, `llm` CLI -- calls the shim via ISO_C_BINDING
  tools/  synthetic GGUF writer/reader (there is no model file offline)
  llmk.py ctypes binding of the same C-ABI, used by tests/ and bench.py
"""
