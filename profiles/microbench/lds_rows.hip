// Microbenchmark: ds_read_b128 of table rows selected per lane (the access shape of the
// bit-sliced scan): which row->bank layouts are conflict-free when lanes pick rows at random?
#include <hip/hip_runtime.h>
#include <cstdio>

template<int MODE>
__global__ __launch_bounds__(256) void k_lds(uint32_t *out, int iters)
{
	__shared__ __attribute__((aligned(16))) uint32_t lds[8192];
	for(int i = threadIdx.x;i < 8192;i += 256) lds[i] = i*2654435761u;
	__syncthreads();
	uint32_t acc = 0;
	uint32_t x = threadIdx.x*2654435761u + blockIdx.x*40503u + 12345u;
	for(int it = 0;it < iters;++it){
		x = x*1664525u + 1013904223u;
		uint32_t bits = x >> 8;
#pragma unroll
		for(int k = 0;k < 8;++k){
			const uint32_t r = bits & 3; bits >>= 2;
			uint32_t off;
			if(MODE == 0) off = (1u << r)*32;            // rows 1,2,4,8 of a 16-row x 32 B table (current layout)
			else if(MODE == 1) off = r*32;               // 4 dense rows of 32 B
			else if(MODE == 2) off = r*64;               // rows 64 B apart
			else if(MODE == 3) off = (lds[0] & 0) + 0;   // one row: pure broadcast
			else off = r*16 + ((threadIdx.x >> 4) & 3)*0; // 4 rows of 16 B
			const uint4 v = *(const uint4 *)((const char *)lds + off + k*512 + (it & 1)*16);
			acc ^= v.x ^ v.y ^ v.z ^ v.w;
		}
	}
	out[blockIdx.x*blockDim.x + threadIdx.x] = acc;
}

int main()
{
	uint32_t *d; hipMalloc(&d, 1 << 26);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
	const int cus = p.multiProcessorCount;
	const double ghz = p.clockRate/1e6;
	const int iters = 4000, wg = 4;
	const char *names[] = {"rows {1,2,4,8}*32B random", "rows {0..3}*32B random", "rows {0..3}*64B random", "single row (broadcast)", "rows {0..3}*16B random"};
	for(int mode = 0;mode < 5;++mode){
		float ms = 0;
		for(int rep = 0;rep < 2;++rep){
			hipEventRecord(e0);
			switch(mode){
				case 0: hipLaunchKernelGGL(k_lds<0>, dim3(cus*wg), dim3(256), 0, 0, d, iters); break;
				case 1: hipLaunchKernelGGL(k_lds<1>, dim3(cus*wg), dim3(256), 0, 0, d, iters); break;
				case 2: hipLaunchKernelGGL(k_lds<2>, dim3(cus*wg), dim3(256), 0, 0, d, iters); break;
				case 3: hipLaunchKernelGGL(k_lds<3>, dim3(cus*wg), dim3(256), 0, 0, d, iters); break;
				default: hipLaunchKernelGGL(k_lds<4>, dim3(cus*wg), dim3(256), 0, 0, d, iters); break;
			}
			hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
		}
		printf("%-28s: %.3f ms -> %.2f cycles per ds_read_b128 wave-instr per CU\n", names[mode], ms, ms*1e-3*ghz*1e9/((double)wg*4*iters*8));
	}
	return 0;
}
