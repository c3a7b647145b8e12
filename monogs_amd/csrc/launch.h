// Kernel launch helper with optional per-kernel hipEvent timing (mgs_profile_*).
#pragma once
#include <hip/hip_runtime.h>

namespace mgs {

bool profile_on();
void profile_push(const char* name, hipEvent_t a, hipEvent_t b);

template <typename K, typename... A>
inline void launch_smem(const char* name, K kernel, dim3 grid, dim3 block, size_t smem,
                        hipStream_t st, A... args) {
  if (profile_on()) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    (void)hipEventRecord(a, st);
    hipLaunchKernelGGL(kernel, grid, block, smem, st, args...);
    (void)hipEventRecord(b, st);
    profile_push(name, a, b);
  } else {
    hipLaunchKernelGGL(kernel, grid, block, smem, st, args...);
  }
}

template <typename K, typename... A>
inline void launch(const char* name, K kernel, dim3 grid, dim3 block, hipStream_t st, A... args) {
  launch_smem(name, kernel, grid, block, 0, st, args...);
}

}  // namespace mgs
