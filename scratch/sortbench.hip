#include <hip/hip_runtime.h>
#include <cstring>
#include <cstdio>
#include <vector>
#include <random>
#include <rocprim/device/device_radix_sort.hpp>
int main(){
  const size_t n=1000000; std::vector<unsigned> h(n); std::mt19937 g(1); for(auto&x:h) x=g();
  unsigned *ki,*ko,*vi,*vo; hipMalloc(&ki,n*4);hipMalloc(&ko,n*4);hipMalloc(&vi,n*4);hipMalloc(&vo,n*4);
  hipMemcpy(ki,h.data(),n*4,hipMemcpyHostToDevice);
  for(unsigned bits: {8u,12u,14u,16u,20u,21u,24u,32u}){
    size_t tb=0; rocprim::radix_sort_pairs(nullptr,tb,ki,ko,vi,vo,n,0,bits,0); void* tmp; hipMalloc(&tmp,tb);
    hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b);
    rocprim::radix_sort_pairs(tmp,tb,ki,ko,vi,vo,n,0,bits,0); hipDeviceSynchronize();
    hipEventRecord(a); for(int i=0;i<20;i++) rocprim::radix_sort_pairs(tmp,tb,ki,ko,vi,vo,n,0,bits,0); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms,a,b); printf("bits %2u: %.1f us (tmp %zu KB)\n",bits,ms/20*1e3,tb/1024); hipFree(tmp);
  }
  return 0;
}
