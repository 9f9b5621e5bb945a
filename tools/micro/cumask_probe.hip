// Does a CU mask that enables only the CUs of SOME XCDs keep a stream's workgroups on those XCDs?
// Hypothesis (from the decode engine's census: a 64-CU mask gave 8 CUs on every XCD): mask bit i = CU (i / 8) of XCD (i % 8).
// The probe builds the mask for "XCDs 4..7 only" under that hypothesis, launches a census kernel and polls (never blocks).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <unistd.h>
#include <vector>
__global__ void census(unsigned* cnt) {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    if (threadIdx.x == 0) atomicAdd(cnt + (x & 15u), 1u);
}
int main(int argc, char** argv) {
    // cumask_probe <xcd_bits>        strided mask: bit i set when XCD (i % 8) is in xcd_bits  (0xF0 = "XCDs 4..7 only")
    // cumask_probe contig <n>        contiguous mask: bits 0 .. n-1
    const bool contig = argc > 2 && !strcmp(argv[1], "contig");
    const unsigned xcd_bits = contig ? strtoul(argv[2], nullptr, 0) : (argc > 1 ? strtoul(argv[1], nullptr, 0) : 0xF0u);
    std::vector<uint32_t> mask(8, 0u);
    for (int i = 0; i < 256; ++i)
        if (contig ? (unsigned)i < xcd_bits : ((xcd_bits >> (i % 8)) & 1u)) mask[i / 32] |= 1u << (i % 32);
    printf("%s mask, argument %u: ", contig ? "contiguous" : "strided", xcd_bits);
    hipStream_t s;
    if (hipExtStreamCreateWithCUMask(&s, 8, mask.data()) != hipSuccess) { printf("mask refused\n"); return 2; }
    unsigned* d;
    hipMalloc(&d, 64);
    hipMemset(d, 0, 64);
    hipEvent_t ev;
    hipEventCreate(&ev);
    hipLaunchKernelGGL(census, dim3(1024), dim3(64), 0, s, d);
    hipEventRecord(ev, s);
    for (int i = 0; i < 300; ++i) {
        if (hipEventQuery(ev) == hipSuccess) {
            unsigned h[16];
            hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
            printf("xcd_bits 0x%02x -> workgroups per XCC:", xcd_bits);
            for (int k = 0; k < 8; ++k) printf(" %u", h[k]);
            printf("\n");
            return 0;
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(10));
    }
    printf("xcd_bits 0x%02x -> NOT FINISHED after 3 s (workgroups routed to an XCD without enabled CUs?)\n", xcd_bits);
    fflush(stdout);
    _exit(1);
}
