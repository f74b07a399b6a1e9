// Launch wrapper of k_block_solve (block_solve.hip): every pointer of the tables is device-visible memory (HBM or pinned, device-mapped).
#pragma once
#include <hip/hip_runtime.h>

#include "block_core.h"

namespace hqblock {

hipError_t block_solve(const ColTable &ct, const ClassTable &cl, const Output &out, uint32_t budget, hipStream_t s);

}  // namespace hqblock
