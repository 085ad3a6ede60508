// TEST INFRASTRUCTURE ONLY: empty stand-in (see container/tensor.h).
#pragma once
