// Microbenchmark: issue rate of the integer/bitwise VALU ops the scan kernels are made of,
// and of conflict-free ds_read_b128, on gfx950.  Prints cycles per wave64 instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template<int OP>
__global__ __launch_bounds__(256) void k_valu(uint32_t *out, int iters, uint32_t seed)
{
	uint32_t r[16];
#pragma unroll
	for(int i = 0;i < 16;++i) r[i] = seed*(i + 1) + threadIdx.x;
	for(int it = 0;it < iters;++it){
#pragma unroll
		for(int i = 0;i < 16;++i){
			const uint32_t a = r[i], b = r[(i + 5) & 15], c = r[(i + 11) & 15];
			if(OP == 0) r[i] = __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
			else if(OP == 1) r[i] = (a & b) | c;                 // v_and_or_b32
			else if(OP == 2) r[i] = a + b;                       // v_add_u32
			else if(OP == 3) r[i] = __popc(a) + b;               // v_bcnt_u32_b32
			else if(OP == 4) r[i] = __builtin_amdgcn_alignbit(a, b, c);
			else if(OP == 5) r[i] = a ^ b;                       // v_xor_b32 (VOP2)
		}
	}
	uint32_t s = 0;
#pragma unroll
	for(int i = 0;i < 16;++i) s ^= r[i];
	out[blockIdx.x*blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_lds(uint32_t *out, int iters, int stride_bytes)
{
	__shared__ __attribute__((aligned(16))) uint32_t lds[4096];
	for(int i = threadIdx.x;i < 4096;i += 256) lds[i] = i*2654435761u;
	__syncthreads();
	uint32_t acc = 0;
	// address pattern: each lane picks one of 4 rows (like A/C/G/T rows of the scan table)
	const uint32_t row = (threadIdx.x*7 + (threadIdx.x >> 3)) & 3;
	const char *base = (const char *)lds + (1u << row)*stride_bytes;
	for(int it = 0;it < iters;++it){
#pragma unroll
		for(int k = 0;k < 8;++k){
			const uint4 v = *(const uint4 *)(base + k*512 + (it & 1)*16);
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
	printf("device %s CUs %d clock %.2f GHz\n", p.gcnArchName, cus, ghz);
	const int iters = 4000;
	const char *names[] = {"v_bitop3_b32", "v_and_or_b32", "v_add_u32", "v_bcnt_u32_b32", "v_alignbit_b32", "v_xor_b32"};
	for(int wg_per_cu = 1;wg_per_cu <= 4;wg_per_cu *= 2){
		for(int op = 0;op < 6;++op){
			float ms = 0;
			for(int rep = 0;rep < 2;++rep){
				hipEventRecord(e0);
				const dim3 g(cus*wg_per_cu), b(256);
				switch(op){
					case 0: hipLaunchKernelGGL(k_valu<0>, g, b, 0, 0, d, iters, 3u); break;
					case 1: hipLaunchKernelGGL(k_valu<1>, g, b, 0, 0, d, iters, 3u); break;
					case 2: hipLaunchKernelGGL(k_valu<2>, g, b, 0, 0, d, iters, 3u); break;
					case 3: hipLaunchKernelGGL(k_valu<3>, g, b, 0, 0, d, iters, 3u); break;
					case 4: hipLaunchKernelGGL(k_valu<4>, g, b, 0, 0, d, iters, 3u); break;
					default: hipLaunchKernelGGL(k_valu<5>, g, b, 0, 0, d, iters, 3u); break;
				}
				hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
			}
			// each SIMD runs wg_per_cu waves, each issuing iters*16 instructions
			const double instr_per_simd = (double)wg_per_cu*iters*16;
			printf("%-16s waves/SIMD %d : %.3f ms  -> %.2f cycles per wave64 instr per SIMD (at %.2f GHz)\n", names[op], wg_per_cu, ms,
				ms*1e-3*ghz*1e9/instr_per_simd, ghz);
		}
	}
	for(int wg_per_cu = 1;wg_per_cu <= 4;wg_per_cu *= 2){
		for(int stride = 32;stride <= 64;stride *= 2){
			float ms = 0;
			for(int rep = 0;rep < 2;++rep){
				hipEventRecord(e0);
				hipLaunchKernelGGL(k_lds, dim3(cus*wg_per_cu), dim3(256), 0, 0, d, iters, stride);
				hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
			}
			const double instr_per_cu = (double)wg_per_cu*4*iters*8;
			printf("ds_read_b128 rows at %d B stride, WG/CU %d : %.3f ms -> %.2f cycles per wave instr per CU\n", stride, wg_per_cu, ms,
				ms*1e-3*ghz*1e9/instr_per_cu);
		}
	}
	return 0;
}
