// profiles/exp/pcie_probe.cpp — how fast can a pageable std::vector-sized buffer cross PCIe in both directions at once?
// (design input for the pipelined host entry points; hipcc -O2 pcie_probe.cpp -o pcie_probe)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    const size_t NI = 199065600, NO = 186900534;
    std::vector<uint8_t> in(NI, 1), out(NO, 2);
    void *di, *dout; CK(hipMalloc(&di, NI)); CK(hipMalloc(&dout, NO));
    hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    for (int rep = 0; rep < 3; ++rep) {
        double t0 = now(); CK(hipMemcpyAsync(di, in.data(), NI, hipMemcpyHostToDevice, s1)); CK(hipStreamSynchronize(s1)); double t1 = now();
        CK(hipMemcpyAsync(out.data(), dout, NO, hipMemcpyDeviceToHost, s2)); CK(hipStreamSynchronize(s2)); double t2 = now();
        printf("serial pageable: H2D %.2f ms (%.1f GB/s)  D2H %.2f ms (%.1f GB/s)\n", (t1 - t0) * 1e3, NI / (t1 - t0) / 1e9, (t2 - t1) * 1e3, NO / (t2 - t1) / 1e9);
    }
    for (int rep = 0; rep < 3; ++rep) {   // both directions at once, two host threads
        double t0 = now();
        std::thread a([&] { (void)hipMemcpyAsync(di, in.data(), NI, hipMemcpyHostToDevice, s1); (void)hipStreamSynchronize(s1); });
        std::thread b([&] { (void)hipMemcpyAsync(out.data(), dout, NO, hipMemcpyDeviceToHost, s2); (void)hipStreamSynchronize(s2); });
        a.join(); b.join(); double t1 = now();
        printf("concurrent pageable (2 threads): %.2f ms\n", (t1 - t0) * 1e3);
    }
    for (int chunks : {4, 8, 16}) {       // chunked, two threads, as a pipeline would issue them
        double t0 = now();
        std::thread a([&] { for (int c = 0; c < chunks; ++c) { size_t o = NI / chunks * c, n = c == chunks - 1 ? NI - o : NI / chunks; (void)hipMemcpyAsync((char*)di + o, in.data() + o, n, hipMemcpyHostToDevice, s1); } (void)hipStreamSynchronize(s1); });
        std::thread b([&] { for (int c = 0; c < chunks; ++c) { size_t o = NO / chunks * c, n = c == chunks - 1 ? NO - o : NO / chunks; (void)hipMemcpyAsync(out.data() + o, (char*)dout + o, n, hipMemcpyDeviceToHost, s2); } (void)hipStreamSynchronize(s2); });
        a.join(); b.join(); double t1 = now();
        printf("concurrent pageable, %d chunks each: %.2f ms\n", chunks, (t1 - t0) * 1e3);
    }
    {   // register in place
        double t0 = now(); CK(hipHostRegister(in.data(), NI, hipHostRegisterDefault)); double t1 = now(); CK(hipHostRegister(out.data(), NO, hipHostRegisterDefault)); double t2 = now();
        printf("hipHostRegister: in %.2f ms, out %.2f ms\n", (t1 - t0) * 1e3, (t2 - t1) * 1e3);
        for (int rep = 0; rep < 2; ++rep) {
            double a0 = now();
            CK(hipMemcpyAsync(di, in.data(), NI, hipMemcpyHostToDevice, s1)); CK(hipMemcpyAsync(out.data(), dout, NO, hipMemcpyDeviceToHost, s2));
            CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2)); double a1 = now();
            printf("concurrent registered: %.2f ms\n", (a1 - a0) * 1e3);
        }
        double u0 = now(); CK(hipHostUnregister(in.data())); CK(hipHostUnregister(out.data())); double u1 = now();
        printf("hipHostUnregister both: %.2f ms\n", (u1 - u0) * 1e3);
    }
    {   // pinned staging + CPU memcpy threads
        void *pi, *po; CK(hipHostMalloc(&pi, NI)); CK(hipHostMalloc(&po, NO));
        for (int th : {1, 4, 8}) {
            double t0 = now();
            std::vector<std::thread> ts;
            for (int t = 0; t < th; ++t) ts.emplace_back([&, t] { size_t o = NI / th * t, n = t == th - 1 ? NI - o : NI / th; memcpy((char*)pi + o, in.data() + o, n); });
            for (auto& t : ts) t.join();
            double t1 = now();
            printf("CPU memcpy pageable->pinned, %d threads: %.2f ms (%.1f GB/s)\n", th, (t1 - t0) * 1e3, NI / (t1 - t0) / 1e9);
        }
        double a0 = now(); CK(hipMemcpyAsync(di, pi, NI, hipMemcpyHostToDevice, s1)); CK(hipMemcpyAsync(po, dout, NO, hipMemcpyDeviceToHost, s2)); CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2)); double a1 = now();
        printf("concurrent pinned both ways: %.2f ms\n", (a1 - a0) * 1e3);
    }
    return 0;
}
