// oracle/ref_stub.cpp -- TEST INFRASTRUCTURE.  Not reference code.
//
// /root/reference/models/index_max_ext/index_max.cpp:124-130 forward-declares two functions that
// live in index_max_cuda.cu (CUDA; not buildable here).  oracle/build_ref.py compiles the
// reference's index_max.cpp *where it lies* together with this stub so that its CPU entry points
// (forward_cpu :73-112, forward_multi_thread_cpu :33-70) can be used as the oracle of record.
// The two device entry points only throw.
#include <torch/extension.h>
#include <stdexcept>

torch::Tensor index_max_forward_cuda(const torch::Tensor, const torch::Tensor, const int) {
    throw std::runtime_error("oracle/_ref: reference CUDA path is not built (CPU oracle only)");
}

torch::Tensor index_max_forward_cuda_shared_mem(const torch::Tensor, const torch::Tensor, const int) {
    throw std::runtime_error("oracle/_ref: reference CUDA path is not built (CPU oracle only)");
}
