// raw buffer store on gfx950: does the record flag word work, are out-of-range lanes dropped, does soffset enter the range check?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(float* o, unsigned int num_records, int sj) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(o, 0, num_records, 0x00020000);
  const unsigned int vo = threadIdx.x < 10 ? threadIdx.x * 4u : 0x80000000u;
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(100.f + threadIdx.x), r, vo, __builtin_amdgcn_readfirstlane(sj) * 40, 0);
}
int main() {
  const int n = 1 << 20;
  float* d; if (hipMalloc(&d, sizeof(float) * n) != hipSuccess) return 1;
  std::vector<float> h(n);
  struct { unsigned int nr; int sj; const char* what; } cases[] = {
    {0x80000000u, 3, "num_records 2^31, record 3"},
    {40u, 0, "num_records 40, record 0"},
    {40u, 3, "num_records 40, record 3 (soffset 120: dropped if soffset is range-checked)"},
    {200u, 3, "num_records 200, record 3 (120 + 40 <= 200)"},
    {130u, 3, "num_records 130, record 3 (only 10 B past soffset)"},
  };
  for (auto& c : cases) {
    (void)hipMemset(d, 0, sizeof(float) * n);
    k<<<1, 64>>>(d, c.nr, c.sj);
    (void)hipMemcpy(h.data(), d, sizeof(float) * n, hipMemcpyDeviceToHost);
    int nz = 0, first = -1, last = -1;
    for (int i = 0; i < n; i++) if (h[i] != 0.f) { nz++; if (first < 0) first = i; last = i; }
    printf("%-80s: %d dwords written, first %d (%.0f) last %d (%.0f)\n", c.what, nz, first, first >= 0 ? h[first] : 0.f, last, last >= 0 ? h[last] : 0.f);
  }
  return 0;
}
