// TEST INFRASTRUCTURE ONLY.  The reference includes <dlpack/dlpack.h>; the same header ships with
// PyTorch as ATen/dlpack.h (build_ref.sh adds torch's include directory).
#pragma once
#include <ATen/dlpack.h>
