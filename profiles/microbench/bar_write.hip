// Can the host write device memory directly (large BAR), so that per-pass tables need no staging kernel?  A ring of four
// 32 KB slots in fine-grained device memory is rewritten by CPU stores (+ sfence) while earlier kernels are still in flight,
// every kernel sums its slot; 4 000 rounds, all sums checked at the end.
// hipcc --offload-arch=gfx950 -O2 -o bar_write bar_write.hip && ./bar_write
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <chrono>
#include <vector>
__global__ void k_sum(const unsigned *p, unsigned n, unsigned *out)
{
	__shared__ unsigned acc;
	if(threadIdx.x == 0) acc = 0;
	__syncthreads();
	unsigned s = 0;
	for(unsigned i = threadIdx.x;i < n;i += blockDim.x) s += p[i];
	atomicAdd(&acc, s);
	__syncthreads();
	if(threadIdx.x == 0) out[blockIdx.x] = acc;
}
int main()
{
	const unsigned n = 8192, ROUNDS = 4000, RING = 4, WGS = 256;
	unsigned *ring = nullptr, *out = nullptr;
	hipError_t e = hipExtMallocWithFlags((void **)&ring, RING*n*4, hipDeviceMallocFinegrained);
	printf("hipExtMallocWithFlags(finegrained): %s\n", hipGetErrorString(e));
	if(e != hipSuccess) return 1;
	hipMalloc((void **)&out, (size_t)ROUNDS*WGS*4);
	hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
	std::vector<unsigned> host(n), want(ROUNDS);
	std::vector<hipEvent_t> done(RING);
	for(auto &ev : done) hipEventCreateWithFlags(&ev, hipEventDisableTiming);
	unsigned x = 12345;
	auto t0 = std::chrono::steady_clock::now();
	for(unsigned r = 0;r < ROUNDS;++r){
		const unsigned slot = r % RING;
		if(r >= RING) hipEventSynchronize(done[slot]);          // the kernel that read this slot four rounds ago is over
		unsigned sum = 0;
		for(unsigned i = 0;i < n;++i){ x = x*1664525u + 1013904223u; host[i] = x >> 8; sum += host[i]; }
		want[r] = sum;
		memcpy(ring + (size_t)slot*n, host.data(), n*4);        // CPU stores straight into device memory
		__builtin_ia32_sfence();
		hipLaunchKernelGGL(k_sum, dim3(WGS), dim3(256), 0, st, ring + (size_t)slot*n, n, out + (size_t)r*WGS);   // every workgroup reads the whole slot
		hipEventRecord(done[slot], st);
	}
	hipStreamSynchronize(st);
	const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
	std::vector<unsigned> got((size_t)ROUNDS*WGS);
	hipMemcpy(got.data(), out, got.size()*4, hipMemcpyDeviceToHost);
	unsigned bad = 0;
	for(unsigned r = 0;r < ROUNDS;++r) for(unsigned w = 0;w < WGS;++w) bad += got[(size_t)r*WGS + w] != want[r];
	printf("%u rounds x %u workgroups in %.1f ms: %u wrong sums\n", ROUNDS, WGS, ms, bad);
	return bad ? 2 : 0;
}
