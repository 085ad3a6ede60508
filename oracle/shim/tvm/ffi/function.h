// TEST INFRASTRUCTURE ONLY: the export macro of tvm-ffi, reduced to a plain C entry point
//   int64_t ref_<name>(const DLTensor* a, const DLTensor* b, char* err, size_t err_len)
// that returns the function's value, or -1 with the PanicError text in `err`.
#pragma once
#include <tvm/ffi/container/tensor.h>

#include <cstring>
#include <exception>

#define TVM_FFI_DLL_EXPORT_TYPED_FUNC(name, fn)                                                    \
  extern "C" __attribute__((visibility("default"))) int64_t ref_##name(                            \
      const DLTensor *a, const DLTensor *b, char *err, size_t err_len) {                           \
    try {                                                                                          \
      return static_cast<int64_t>(fn(tvm::ffi::TensorView(a), tvm::ffi::TensorView(b)));           \
    } catch (const std::exception &e) {                                                            \
      if (err && err_len) {                                                                        \
        std::strncpy(err, e.what(), err_len - 1);                                                  \
        err[err_len - 1] = 0;                                                                      \
      }                                                                                            \
      return -1;                                                                                   \
    }                                                                                              \
  }
