/* cuda_on_cpu/cuda.h -- TEST INFRASTRUCTURE ONLY; see cuda_runtime.h */
#include "cuda_runtime.h"
