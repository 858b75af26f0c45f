// Umbrella include (kept for tools/): see zk_ntt_kernels.h and zk_msm_kernels.h.
#pragma once
#include "zk_ntt_kernels.h"
#include "zk_msm_kernels.h"
