// TEST INFRASTRUCTURE ONLY -- a stand-in for the handful of apache-tvm-ffi declarations that
// /root/reference/python/minisgl/kernel/csrc/src/radix.cpp uses (tvm_ffi is not installed and there
// is no network), so that the reference's OWN radix.cpp compiles unmodified into oracle/_ref/.
// Only the members radix.cpp:12-40 touches exist: a non-owning view over a DLTensor.
#pragma once
#include <dlpack/dlpack.h>

#include <cstddef>
#include <cstdint>

inline bool operator==(const DLDataType &a, const DLDataType &b) {
  return a.code == b.code && a.bits == b.bits && a.lanes == b.lanes;
}

namespace tvm::ffi {

class TensorView {
public:
  explicit TensorView(const DLTensor *t) : t_(t) {}
  int ndim() const { return t_->ndim; }
  bool is_contiguous() const {
    if (t_->strides == nullptr) return true;
    int64_t expect = 1;
    for (int i = t_->ndim - 1; i >= 0; --i) {
      if (t_->shape[i] != 1 && t_->strides[i] != expect) return false;
      expect *= t_->shape[i];
    }
    return true;
  }
  DLDevice device() const { return t_->device; }
  DLDataType dtype() const { return t_->dtype; }
  const void *data_ptr() const { return static_cast<const char *>(t_->data) + t_->byte_offset; }
  int64_t size(int i) const { return t_->shape[i]; }

private:
  const DLTensor *t_;
};

} // namespace tvm::ffi
