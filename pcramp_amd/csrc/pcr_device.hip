// MI355X (gfx950) implementation of the C-ABI in include/pcramp_hip.h.
//
// HBM layout (per sequence set)
//   planes[]  : one uint4 {A,C,G,T} per 32-base block; bit j of .x = base 32*b+j may be 'A', ...
//               (the reference's 4-bit IUPAC nibble, bit-transposed: same 4 bits/base as
//               Sequence::seq_buffer, sequence.h:86).  EOS / past-the-end = no bit in any plane.
//               Each sequence owns ceil(L/32)+2 blocks (two zero halo blocks).
//   nib[]     : the same codes as 4 u32 per block, nibble j%8 of word j/8 = base 32*b+j (table index of
//               the bit-sliced scan, pcr_scan_bitsliced.inc).
//   tb[]      : 2 bits per base (A,C,G,T = 0..3, anything else 0), 2 u32 per block: q-gram codes of the seed scan.
//   valid[]   : one u32 per block; bit j = the 32-base window starting at 32*b+j is a REGULAR
//               window that Sequence::pack emits (sequence.cpp:127-153 filters applied).
//   irr[]     : explicit irregular words (pcr_host.hpp); irr_scan[]: the same in scan order as 2-bit codes (k_seed2).
//   tile_desc[]: one 16-byte descriptor per 1024-window tile (k_seed2).
// Kernels
//   k_transpose, k_valid, k_tile_degen, k_tile_desc : load-time index build (replaces Sequence::operator= + the filter half of pack)
//   k_stage                : per-pass tables out of host-mapped memory + clearing of the pass's control block
//   k_seed2                : the default match scan: seed filter on 9-grams, tables built in LDS (pcr_scan_seed2.inc)
//   k_seed (+ k_seed_tables, scan_irr_block), k_scan2, k_scan, k_scan_irr :
//                            the other forms of the oligo x window match scan + per-(sequence,candidate) running max
//                            (select_words.cpp:88-117 over the implicit word index); pcr_scan_seed.inc,
//                            pcr_scan_bitsliced.inc
//   k_touched, k_finalize, k_post : per-sequence arg-max-with-ties filter, sort, dedupe -> the device word DB
//                            (select_words.cpp:100-138); k_post = the whole tail of the fused pass in one launch
//   k_match, k_pair        : match_words / find_oligo_match / find_amplicon_match / update_identity /
//                            sqrtf(f*r) test (optimize.cpp:209-301, pcr_assay.cpp:12-69,338-441,544-578)
//   k_pair_moves, k_pair_moves_batch : the same sweep for the trial words of a local-search move (optimize_pcr.cpp; pcr_optimize.inc)
//   k_sw, k_sw_words, k_bg_*, k_mx_* : SeqOverlap Smith-Waterman and the background / multiplex screens (pcr_sw.inc)
//   thermo::k_thermo_wave  : NucCruc (pcr_thermo.inc)
// Host side in the same library: pcr_optimize.inc (optimize() batched over trial assays), pcr_sampler.inc, pcr_multiplex.inc,
// pcr_multiplex_screen.inc, pcr_writers.inc; pcr_exchange.inc: the bitset all-gather over RCCL.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include <stdint.h>
#include <string.h>
#include <math.h>
#include <stdlib.h>
#include <string>
#include <vector>
#include <chrono>
#include <algorithm>
#include <unordered_map>
#include <type_traits>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <functional>
#include <memory>

#include "../../include/pcramp_hip.h"
#include "pcr_host.hpp"

using pcrhost::Planes;

namespace {

thread_local std::string g_err;

#define HIP_TRY(expr) do{ hipError_t e_ = (expr); if(e_ != hipSuccess){ \
	g_err = std::string(#expr) + ": " + hipGetErrorString(e_); return PCR_ERR_DEVICE; } }while(0)

constexpr int TILE_POS = 1024;       // window starts per workgroup
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_NPOS = TILE_POS/SCAN_THREADS;

// dedupe / sort key of a DB entry: [seq:24][loc+64:32][strand-1:1][kind:1][ord:6]
constexpr int KEY_SEQ_SHIFT = 40;
constexpr int KEY_LOC_SHIFT = 8;
constexpr int LOC_BIAS = 64;

struct Hit { uint64_t key; uint32_t cand; uint32_t cnt; };

// host-mapped mailbox slot: a pass's counters, then its sequence number (release store)
struct PassMail { volatile uint32_t counters[4]; volatile uint32_t seq; uint32_t pad[11]; };

constexpr uint32_t N_TOUCHED_UNKNOWN_DEV = 0xFFFFFFFFu;   // published instead of a touched count by the fused tail

struct IrrDev { Planes w; int32_t loc; uint32_t seq; uint32_t meta; /* strand | cws<<8 | ord<<16 */ uint32_t local_id; };

struct DevEntry { Planes w; int32_t loc; uint32_t seq; uint32_t strand; uint32_t pad; };

struct OligoDev {        // one assay oligo for the amplicon screen
	Planes m;
	uint32_t floor2;     // unsigned(size * thr^2), optimize.cpp:293
	float norm;          // float(1.0/len), optimize.cpp:221
	int32_t start, stop; // Word::start()/stop()
	uint32_t p1, p2;     // last two primer bases (optimize.cpp:236)
};

__constant__ float c_taq_mama[256];

// word.cpp:252-272 -- Table 2 of Li et al., Genomics 83 (2004) 311-320 (rows: template pair,
// columns: primer pair, order {CC,GC,AC,TC,CG,GG,AG,TG,CA,GA,AA,TA,CT,GT,AT,TT}).
const float h_taq_mama[256] = {
	1.000f, 0.968f, 0.947f, 1.034f, 0.547f, 0.253f, 0.230f, 0.359f, 0.606f, 0.282f, 0.372f, 0.347f, 0.957f, 0.382f, 0.399f, 0.687f,
	0.989f, 1.000f, 1.023f, 1.000f, 0.420f, 0.662f, 0.445f, 0.367f, 0.870f, 0.512f, 0.492f, 0.508f, 0.372f, 1.000f, 0.492f, 0.714f,
	1.011f, 1.000f, 1.000f, 1.000f, 0.459f, 0.277f, 0.570f, 0.343f, 0.927f, 0.362f, 0.590f, 0.542f, 0.439f, 0.488f, 0.978f, 0.662f,
	1.000f, 0.907f, 1.000f, 1.000f, 0.382f, 0.234f, 0.228f, 0.542f, 0.763f, 0.309f, 0.410f, 0.473f, 0.426f, 0.347f, 0.423f, 0.947f,
	0.590f, 0.334f, 0.445f, 0.323f, 1.000f, 0.978f, 0.927f, 0.989f, 0.907f, 0.645f, 0.525f, 0.455f, 0.927f, 0.408f, 0.408f, 0.707f,
	0.327f, 0.595f, 0.319f, 0.396f, 0.947f, 1.000f, 0.978f, 0.989f, 0.405f, 0.861f, 0.681f, 0.512f, 0.410f, 0.968f, 0.452f, 0.714f,
	0.410f, 0.420f, 0.590f, 0.311f, 1.023f, 1.000f, 1.000f, 1.000f, 0.488f, 0.898f, 0.907f, 0.566f, 0.442f, 0.449f, 0.989f, 0.707f,
	0.423f, 0.343f, 0.305f, 0.585f, 1.034f, 0.879f, 0.927f, 1.000f, 0.473f, 0.720f, 0.547f, 0.957f, 0.459f, 0.374f, 0.459f, 1.023f,
	1.023f, 0.429f, 0.473f, 0.477f, 1.023f, 0.466f, 0.420f, 0.477f, 1.000f, 0.978f, 0.907f, 0.978f, 0.907f, 0.380f, 0.525f, 0.669f,
	0.442f, 1.046f, 0.455f, 0.470f, 0.432f, 1.058f, 0.481f, 0.485f, 0.917f, 1.000f, 1.023f, 1.023f, 0.336f, 0.968f, 0.534f, 0.639f,
	0.617f, 0.452f, 1.011f, 0.439f, 0.492f, 0.504f, 0.978f, 0.462f, 0.989f, 0.947f, 1.000f, 0.978f, 0.405f, 0.405f, 0.888f, 0.606f,
	0.601f, 0.377f, 0.377f, 1.046f, 0.500f, 0.399f, 0.408f, 1.034f, 0.978f, 0.720f, 0.870f, 1.000f, 0.402f, 0.313f, 0.651f, 0.927f,
	0.978f, 0.462f, 0.466f, 0.488f, 0.420f, 0.239f, 0.225f, 0.336f, 0.504f, 0.269f, 0.319f, 0.656f, 1.000f, 0.835f, 0.907f, 1.034f,
	0.429f, 1.011f, 0.473f, 0.477f, 0.340f, 0.413f, 0.357f, 0.354f, 0.352f, 0.538f, 0.413f, 0.794f, 0.927f, 1.000f, 1.058f, 1.000f,
	0.595f, 0.492f, 0.968f, 0.485f, 0.367f, 0.282f, 0.388f, 0.439f, 0.413f, 0.309f, 0.566f, 0.917f, 0.957f, 0.957f, 1.000f, 0.989f,
	0.590f, 0.380f, 0.410f, 0.968f, 0.364f, 0.223f, 0.230f, 0.416f, 0.321f, 0.239f, 0.301f, 0.645f, 0.978f, 0.714f, 0.947f, 1.000f
};

// ============================================================================== device helpers
__device__ __forceinline__ uint32_t funnel(uint32_t lo, uint32_t hi, uint32_t sh)
{
	return __builtin_amdgcn_alignbit(hi, lo, sh);   // ({hi,lo} >> (sh & 31)) & 0xffffffff
}

__device__ __forceinline__ uint32_t match_count(uint32_t wa, uint32_t wc, uint32_t wg, uint32_t wt, const uint4 m)
{
	// # slots whose base sets intersect == Word::operator& (word.cpp:151-154) in plane form
	return __popc((wa & m.x) | (wc & m.y) | (wg & m.z) | (wt & m.w));
}

// ============================================================================== load-time kernels
// One thread per 32-base block: nibbles (high first, sequence.h:223-228) -> four bit planes.
__global__ void k_transpose(const uint8_t *__restrict__ packed, const uint64_t *__restrict__ byte_off,
	const uint64_t *__restrict__ len, const uint64_t *__restrict__ blk_off, const uint32_t *__restrict__ blk_seq,
	uint4 *__restrict__ planes, uint32_t *__restrict__ nib, uint32_t *__restrict__ tb, uint64_t total_blocks)
{
	const uint64_t gb = (uint64_t)blockIdx.x*blockDim.x + threadIdx.x;
	if(gb >= total_blocks) return;
	const uint32_t s = blk_seq[gb];
	const uint64_t b = gb - blk_off[s];
	const uint64_t L = len[s];
	const uint8_t *src = packed + byte_off[s];
	uint32_t a = 0, c = 0, g = 0, t = 0;
	uint32_t nw[4] = {0, 0, 0, 0};   // the same 32 codes, nibble j%8 of word j/8 (v2 scan's table index)
	uint32_t tw[2] = {0, 0};
	for(int j = 0;j < 32;++j){
		const uint64_t pos = b*32 + j;
		if(pos >= L) break;
		const uint8_t v = src[pos >> 1];
		const uint32_t code = (pos & 1) ? (v & 0xF) : (v >> 4);
		a |= (code & 1u) << j;
		c |= ((code >> 1) & 1u) << j;
		g |= ((code >> 2) & 1u) << j;
		t |= ((code >> 3) & 1u) << j;
		nw[j >> 3] |= code << ((j & 7)*4);
		// 2-bit code for the seed scan: A,C,G,T -> 0..3; anything else (IUPAC, EOS) -> 0
		const uint32_t c2 = (code == 2) ? 1u : (code == 4) ? 2u : (code == 8) ? 3u : 0u;
		tw[j >> 4] |= c2 << ((j & 15)*2);
	}
	planes[gb] = make_uint4(a, c, g, t);
	((uint4 *)nib)[gb] = make_uint4(nw[0], nw[1], nw[2], nw[3]);
	((uint2 *)tb)[gb] = make_uint2(tw[0], tw[1]);
}

// One thread per block: validity of the 32 windows that START in it.  A window is regular iff
// all 32 slots are non-EOS; then pack's filters (GC first, then degeneracy; sequence.cpp:127-153).
__global__ void k_valid(const uint4 *__restrict__ planes, const uint64_t *__restrict__ blk_off,
	const uint32_t *__restrict__ blk_seq, const uint64_t *__restrict__ nblk_real, uint32_t *__restrict__ valid,
	uint64_t total_blocks, uint32_t max_degen, uint64_t gc_ok, uint64_t first_block, uint64_t n_blocks)
{
	const uint64_t i = (uint64_t)blockIdx.x*blockDim.x + threadIdx.x;
	if(i >= n_blocks) return;
	const uint64_t gb = first_block + i;
	if(gb >= total_blocks) return;
	const uint32_t s = blk_seq[gb];
	const uint64_t b = gb - blk_off[s];
	// halo blocks start no window
	if(b >= nblk_real[s]){ valid[gb] = 0; return; }
	const uint4 lo = planes[gb];
	const uint4 hi = planes[gb + 1];   // exists: every sequence has two halo blocks
	// per-position base multiplicity planes
	uint32_t nzl, d2l, d3l, d4l, cgl, nzh, d2h, d3h, d4h, cgh;
	{
		const uint32_t x = lo.x ^ lo.y, y = lo.x & lo.y, u = lo.z ^ lo.w, v = lo.z & lo.w;
		const uint32_t ones = x ^ u, c1 = x & u;
		const uint32_t twos = y ^ v ^ c1;
		const uint32_t fours = (y & v) | (c1 & (y ^ v));
		nzl = lo.x | lo.y | lo.z | lo.w;
		d2l = ~ones & twos & ~fours; d3l = ones & twos; d4l = fours;
		cgl = lo.y | lo.z;
	}
	{
		const uint32_t x = hi.x ^ hi.y, y = hi.x & hi.y, u = hi.z ^ hi.w, v = hi.z & hi.w;
		const uint32_t ones = x ^ u, c1 = x & u;
		const uint32_t twos = y ^ v ^ c1;
		const uint32_t fours = (y & v) | (c1 & (y ^ v));
		nzh = hi.x | hi.y | hi.z | hi.w;
		d2h = ~ones & twos & ~fours; d3h = ones & twos; d4h = fours;
		cgh = hi.y | hi.z;
	}
	uint32_t out = 0;
	for(uint32_t j = 0;j < 32;++j){
		if(funnel(nzl, nzh, j) != 0xFFFFFFFFu) continue;
		const uint32_t ngc = __popc(funnel(cgl, cgh, j));
		if(!((gc_ok >> ngc) & 1ull)) continue;
		const uint32_t n2 = __popc(funnel(d2l, d2h, j));
		const uint32_t n3 = __popc(funnel(d3l, d3h, j));
		const uint32_t n4 = __popc(funnel(d4l, d4h, j));
		// Word::degeneracy() > thr, exact integer form (pcr_host.hpp: degeneracy_exceeds)
		const uint32_t e = n2 + 2*n4;
		bool exceeds;
		if(e >= 33){ exceeds = true; }
		else{
			uint64_t p3 = 1;
			for(uint32_t k = 0;k < n3;++k) p3 *= 3;
			exceeds = p3 > ((uint64_t)max_degen >> e);
		}
		if(!exceeds) out |= (1u << j);
	}
	valid[gb] = out;
}

// ============================================================================== select_words
// Hits are appended to a fixed-capacity bucket per sequence (`cap` slots each), so the arg-max
// filter, the dedupe and the ordering by WordMatch::loc can be done per sequence by one workgroup
// (k_finalize) with no global sort.  counters[0]: overflow flag, counters[2]: largest bucket fill seen.
struct HitSink { uint32_t *best; Hit *hits; uint32_t *seq_count; uint32_t *counters; uint32_t cap; uint32_t ncand; uint32_t epoch; };
// counters[3] / touched[]: the sequences that received at least one hit, listed by k_touched after the scans
// (one wave-aggregated atomic per 64 sequences: appending from record_hit cost ~35 ns per touched sequence,
// all on one address -- more than the whole seed scan).
// best[] holds (pass epoch << 8) | count, so it never needs clearing: values of earlier passes compare lower.

__device__ __forceinline__ void record_hit_inline(const HitSink &k, uint32_t seq, uint32_t cand, uint64_t key, uint32_t cnt)
{
	const uint32_t tagged = (k.epoch << 8) | cnt;
	const uint32_t old = atomicMax(&k.best[(size_t)seq*k.ncand + cand], tagged);
	if(tagged >= old){
		const uint32_t slot = atomicAdd(&k.seq_count[seq], 1u);
		if(slot < k.cap){
			Hit h; h.key = key; h.cand = cand; h.cnt = cnt;
			k.hits[(size_t)seq*k.cap + slot] = h;
		}
		else{ atomicOr(&k.counters[0], 1u); atomicMax(&k.counters[2], slot + 1); }
	}
}

// (by value: a reference would make every kernel spill its HitSink to scratch at entry -- 64 B per lane of HBM writes)
__device__ __noinline__ void record_hit(const HitSink k, uint32_t seq, uint32_t cand, uint64_t key, uint32_t cnt)
{
	record_hit_inline(k, seq, cand, key, cnt);
}

__device__ __forceinline__ uint64_t make_key(uint32_t seq, int32_t loc, uint32_t strand /*1|2*/, uint32_t kind, uint32_t ord)
{
	return ((uint64_t)seq << KEY_SEQ_SHIFT) | ((uint64_t)(uint32_t)(loc + LOC_BIAS) << KEY_LOC_SHIFT) |
		((uint64_t)(strand - 1) << 7) | ((uint64_t)kind << 6) | ord;
}

__global__ void k_touched(const uint32_t *__restrict__ seq_count, uint32_t n, uint32_t *__restrict__ counters, uint32_t *__restrict__ touched)
{
	const uint32_t s = blockIdx.x*blockDim.x + threadIdx.x;
	const uint32_t fill = (s < n) ? seq_count[s] : 0u;
	const bool has = fill > 0;
	const uint64_t mask = __ballot(has);
	if(!mask) return;
	const uint32_t lane = threadIdx.x & 63u, leader = (uint32_t)__builtin_ctzll(mask);
	{   // counters[2]: the largest bucket fill of the pass (the host shrinks oversized buckets before the next pass)
		uint32_t mx = fill;
		for(int d = 32;d > 0;d >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, d));
		if(lane == leader) atomicMax(&counters[2], mx);
	}
	uint32_t base = 0;
	if(lane == leader) base = atomicAdd(&counters[3], (uint32_t)__builtin_popcountll(mask));
	base = __shfl(base, leader);
	if(has) touched[base + (uint32_t)__builtin_popcountll(mask & ((1ull << lane) - 1ull))] = s;
}

// the same list from the DB's segment ends (a sequence with hits has at least one entry), for DBs whose per-sequence fills
// the fused tail has already zeroed
__global__ void k_touched_seg(const uint32_t *__restrict__ seg_hi, uint32_t n, uint32_t *__restrict__ counters, uint32_t *__restrict__ touched)
{
	const uint32_t s = blockIdx.x*blockDim.x + threadIdx.x;
	const bool has = s < n && seg_hi[s] != 0;
	const uint64_t mask = __ballot(has);
	if(!mask) return;
	const uint32_t lane = threadIdx.x & 63u, leader = (uint32_t)__builtin_ctzll(mask);
	uint32_t base = 0;
	if(lane == leader) base = atomicAdd(&counters[3], (uint32_t)__builtin_popcountll(mask));
	base = __shfl(base, leader);
	if(has) touched[base + (uint32_t)__builtin_popcountll(mask & ((1ull << lane) - 1ull))] = s;
}

// v1 match scan: one lane = one window start; the window's four 32-bit plane slices live in
// registers; candidates are wave-uniform (scalar loads), 4 and/or + popcount + compare each.
__global__ __launch_bounds__(SCAN_THREADS) void k_scan(
	const uint4 *__restrict__ planes, const uint32_t *__restrict__ valid,
	const uint64_t *__restrict__ blk_off, const uint64_t *__restrict__ len, const uint8_t *__restrict__ active,
	const uint32_t *__restrict__ tile_seq, const uint32_t *__restrict__ tile_pos0,
	const uint4 *__restrict__ cand_fwd, const uint4 *__restrict__ cand_rc, const uint32_t *__restrict__ cand_floor,
	uint32_t ncand, HitSink sink)
{
	const uint32_t tile = blockIdx.x;
	const uint32_t seq = tile_seq[tile];
	if(!active[seq]) return;
	const uint64_t L = len[seq];
	const uint64_t base = blk_off[seq];
	const uint32_t p0 = tile_pos0[tile];

	uint32_t wa[SCAN_NPOS], wc[SCAN_NPOS], wg[SCAN_NPOS], wt[SCAN_NPOS], ok[SCAN_NPOS];
#pragma unroll
	for(int i = 0;i < SCAN_NPOS;++i){
		const uint32_t p = p0 + threadIdx.x + SCAN_THREADS*i;
		wa[i] = wc[i] = wg[i] = wt[i] = 0; ok[i] = 0;
		if((uint64_t)p + 32 <= L){
			const uint32_t b = p >> 5, sh = p & 31;
			const uint4 lo = planes[base + b];
			const uint4 hi = planes[base + b + 1];
			wa[i] = funnel(lo.x, hi.x, sh); wc[i] = funnel(lo.y, hi.y, sh);
			wg[i] = funnel(lo.z, hi.z, sh); wt[i] = funnel(lo.w, hi.w, sh);
			ok[i] = (valid[base + b] >> sh) & 1u;
		}
	}

	for(uint32_t c = 0;c < ncand;++c){
		const uint4 mf = cand_fwd[c];
		const uint4 mr = cand_rc[c];
		const uint32_t fl = cand_floor[c];
#pragma unroll
		for(int i = 0;i < SCAN_NPOS;++i){
			const uint32_t cf = match_count(wa[i], wc[i], wg[i], wt[i], mf);
			const uint32_t cr = match_count(wa[i], wc[i], wg[i], wt[i], mr);
			if((cf >= fl || cr >= fl) && ok[i]){
				const int32_t p = (int32_t)(p0 + threadIdx.x + SCAN_THREADS*i);
				if(cf >= fl) record_hit(sink, seq, c, make_key(seq, p, 1, 0, 0), cf);       // sequence.cpp:184
				if(cr >= fl) record_hit(sink, seq, c, make_key(seq, p + 31, 2, 0, 0), cr);  // sequence.cpp:190
			}
		}
	}
}

// Irregular words: every lane keeps IRR_PER_LANE words in registers; the candidates are walked in the
// outer loop, so their planes and floors are wave-uniform (scalar loads, SGPR operands): 4 and/or +
// popcount + compare per (word, candidate).  grid.x = ceil(n_live / (IRR_THREADS*IRR_PER_LANE)).
constexpr int IRR_THREADS = 256, IRR_PER_LANE = 2;
struct IrrArgs { const IrrDev *irr; const uint32_t *perm; uint32_t n_live; uint32_t off_mask; /* slot offsets used by forward seeds */ };

// the work of irregular-scan workgroup `block` (IRR_THREADS lanes)
__device__ __forceinline__ void scan_irr_block(uint32_t block, const IrrDev *__restrict__ irr, const uint32_t *__restrict__ perm, uint32_t n_live,
	const uint8_t *__restrict__ active, const uint4 *__restrict__ cand_fwd, const uint32_t *__restrict__ cand_floor, uint32_t ncand,
	const HitSink &sink)
{
	uint32_t wa[IRR_PER_LANE], wc[IRR_PER_LANE], wg[IRR_PER_LANE], wt[IRR_PER_LANE];
	const uint32_t i0 = block*(IRR_THREADS*IRR_PER_LANE) + threadIdx.x;
#pragma unroll
	for(int k = 0;k < IRR_PER_LANE;++k){
		const uint32_t i = i0 + k*IRR_THREADS;
		wa[k] = wc[k] = wg[k] = wt[k] = 0;               // an empty word matches nothing (floors are >= 1 here, see below)
		if(i < n_live){                                  // the words whose size counter reaches min_oligo_length (sequence.cpp:157,239)
			const IrrDev e = irr[perm[i]];
			if(active[e.seq]){ wa[k] = e.w.a; wc[k] = e.w.c; wg[k] = e.w.g; wt[k] = e.w.t; }
		}
	}
#pragma unroll 8
	for(uint32_t c = 0;c < ncand;++c){
		const uint4 m = cand_fwd[c];
		const uint32_t fl = max(cand_floor[c], 1u);      // floor 0 ("everything matches") still needs one matching slot to exist as a word
		const bool zero_floor = cand_floor[c] == 0;
#pragma unroll
		for(int k = 0;k < IRR_PER_LANE;++k){
			const uint32_t cnt = match_count(wa[k], wc[k], wg[k], wt[k], m);
			if(cnt >= fl || (zero_floor && (wa[k] | wc[k] | wg[k] | wt[k]))){
				const IrrDev e = irr[perm[i0 + k*IRR_THREADS]];
				record_hit(sink, e.seq, c, make_key(e.seq, e.loc, e.meta & 0xFF, 1, (e.meta >> 16) & 0xFF), cnt);
			}
		}
	}
}

__global__ __launch_bounds__(IRR_THREADS) void k_scan_irr(const IrrDev *__restrict__ irr, const uint32_t *__restrict__ perm, uint32_t n_live,
	const uint8_t *__restrict__ active, const uint4 *__restrict__ cand_fwd, const uint32_t *__restrict__ cand_floor, uint32_t ncand,
	HitSink sink)
{
	scan_irr_block(blockIdx.x, irr, perm, n_live, active, cand_fwd, cand_floor, ncand, sink);
}

// LDS hand-off between lanes of ONE wave: order the wave's own LDS traffic, no workgroup barrier
__device__ __forceinline__ void wave_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }

#include "pcr_scan_bitsliced.inc"
#include "pcr_scan_seed.inc"
#include "pcr_scan_seed2.inc"
#include "pcr_scan_seed3.inc"

// One workgroup per sequence: keep the hits that attain the final per-(sequence,candidate) maximum
// (select_words.cpp:100-117), sort them by (loc, strand, kind, ord) (bitonic, LDS), drop duplicates
// (the union over candidates, select_words.cpp:126-128) and materialise the DB entries of the
// sequence into its slot range [seq*cap, seq*cap + n).  seg_hi[seq] = seq*cap + n.
constexpr int FIN_WAVES = 8;          // one wave per sequence, 8 sequences per workgroup (1 when the buckets outgrow 1024 hits: LDS)
constexpr uint32_t MAX_BUCKET_CAP = 8192;   // 8192 keys x 8 B = the 64 KB of LDS one wave may sort in


// (seq_base: blk_off of the key's sequence, which callers that work on one sequence have loaded long before)
__device__ __forceinline__ void materialise_entry_at(uint64_t k, const uint4 *__restrict__ planes, uint64_t seq_base,
	const IrrDev *__restrict__ irr, const uint32_t *__restrict__ irr_off, DevEntry &e)
{
	const uint32_t seq = (uint32_t)(k >> KEY_SEQ_SHIFT);
	const int32_t loc = (int32_t)(uint32_t)((k >> KEY_LOC_SHIFT) & 0xFFFFFFFFull) - LOC_BIAS;
	const uint32_t strand = (uint32_t)((k >> 7) & 1) + 1;
	const uint32_t kind = (uint32_t)((k >> 6) & 1), ord = (uint32_t)(k & 63);
	e.loc = loc; e.seq = seq; e.strand = strand; e.pad = 0;
	if(kind == 0){
		const uint32_t p = (strand == 1) ? (uint32_t)loc : (uint32_t)(loc - 31);
		const uint32_t b = p >> 5, sh = p & 31;
		const uint4 lo = planes[seq_base + b];
		const uint4 hi = planes[seq_base + b + 1];
		const uint32_t a = funnel(lo.x, hi.x, sh), c = funnel(lo.y, hi.y, sh);
		const uint32_t g = funnel(lo.z, hi.z, sh), t = funnel(lo.w, hi.w, sh);
		if(strand == 1){ e.w.a = a; e.w.c = c; e.w.g = g; e.w.t = t; }
		else{ e.w.a = __brev(t); e.w.t = __brev(a); e.w.c = __brev(g); e.w.g = __brev(c); }   // Word::complement, word.h:140
	}
	else{
		e.w.a = e.w.c = e.w.g = e.w.t = 0;
		for(uint32_t j = irr_off[seq];j < irr_off[seq + 1];++j){
			const IrrDev r = irr[j];
			if(r.loc == loc && (r.meta & 0xFF) == strand && ((r.meta >> 16) & 0xFF) == ord){ e.w = r.w; break; }
		}
	}
}

__device__ void materialise_entry(uint64_t k, const uint4 *__restrict__ planes, const uint64_t *__restrict__ blk_off,
	const IrrDev *__restrict__ irr, const uint32_t *__restrict__ irr_off, DevEntry &e)
{
	const uint32_t seq = (uint32_t)(k >> KEY_SEQ_SHIFT);
	const int32_t loc = (int32_t)(uint32_t)((k >> KEY_LOC_SHIFT) & 0xFFFFFFFFull) - LOC_BIAS;
	const uint32_t strand = (uint32_t)((k >> 7) & 1) + 1;
	const uint32_t kind = (uint32_t)((k >> 6) & 1), ord = (uint32_t)(k & 63);
	e.loc = loc; e.seq = seq; e.strand = strand; e.pad = 0;
	if(kind == 0){
		const uint32_t p = (strand == 1) ? (uint32_t)loc : (uint32_t)(loc - 31);
		const uint32_t b = p >> 5, sh = p & 31;
		const uint4 lo = planes[blk_off[seq] + b];
		const uint4 hi = planes[blk_off[seq] + b + 1];
		const uint32_t a = funnel(lo.x, hi.x, sh), c = funnel(lo.y, hi.y, sh);
		const uint32_t g = funnel(lo.z, hi.z, sh), t = funnel(lo.w, hi.w, sh);
		if(strand == 1){ e.w.a = a; e.w.c = c; e.w.g = g; e.w.t = t; }
		else{ e.w.a = __brev(t); e.w.t = __brev(a); e.w.c = __brev(g); e.w.g = __brev(c); }   // Word::complement, word.h:140
	}
	else{
		// the irregular word with this (loc, strand, ord) in the sequence's list
		e.w.a = e.w.c = e.w.g = e.w.t = 0;
		for(uint32_t j = irr_off[seq];j < irr_off[seq + 1];++j){
			const IrrDev r = irr[j];
			if(r.loc == loc && (r.meta & 0xFF) == strand && ((r.meta >> 16) & 0xFF) == ord){ e.w = r.w; break; }
		}
	}
}

template<int WAVES>
__global__ __launch_bounds__(64*WAVES) void k_finalize(const Hit *__restrict__ hits, const uint32_t *__restrict__ seq_count,
	uint32_t cap, const uint32_t *__restrict__ best, uint32_t ncand, const uint4 *__restrict__ planes,
	const uint64_t *__restrict__ blk_off, const IrrDev *__restrict__ irr, const uint32_t *__restrict__ irr_off,
	DevEntry *__restrict__ db, uint32_t *__restrict__ seg_hi, uint32_t *__restrict__ counters, uint32_t epoch,
	const uint32_t *__restrict__ touched)
{
	extern __shared__ __attribute__((aligned(16))) uint64_t fin_lds[];   // WAVES x np2cap keys
	__shared__ uint32_t part_all[64*WAVES];
	const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	uint32_t np2cap = 1; while(np2cap < cap) np2cap <<= 1;
	// only sequences that received hits (seg_hi of the others stays 0 = empty).  One workgroup per 8 sequences of
	// the whole set; 256 fat workgroups walking the list were slower (10.7 vs 7.4 us at C2: latency-, not dispatch-bound)
	for(uint32_t t_idx = blockIdx.x*WAVES + wave;t_idx < counters[3];t_idx += gridDim.x*WAVES){
	const uint32_t seq = touched[t_idx];
	uint64_t *fin_keys = fin_lds + (size_t)wave*np2cap;
	uint32_t *part = part_all + wave*64;
	const uint32_t n = min(seq_count[seq], cap);
	uint32_t np2 = 1; while(np2 < n) np2 <<= 1;
	for(uint32_t i = lane;i < np2;i += 64){
		uint64_t k = ~0ull;
		if(i < n){
			const Hit h = hits[(size_t)seq*cap + i];
			if(((epoch << 8) | h.cnt) == best[(size_t)seq*ncand + h.cand]) k = h.key;
		}
		fin_keys[i] = k;
	}
	wave_sync();
	for(uint32_t k = 2;k <= np2;k <<= 1){
		for(uint32_t j = k >> 1;j > 0;j >>= 1){
			for(uint32_t i = lane;i < np2;i += 64){
				const uint32_t l = i ^ j;
				if(l > i){
					const uint64_t a = fin_keys[i], b = fin_keys[l];
					const bool up = ((i & k) == 0);
					if((a > b) == up){ fin_keys[i] = b; fin_keys[l] = a; }
				}
			}
			wave_sync();
		}
	}
	// distinct, non-sentinel keys -> ranks (each lane owns a contiguous chunk)
	const uint32_t chunk = (np2 + 63)/64;
	const uint32_t lo = lane*chunk, hi = min(np2, lo + chunk);
	uint32_t cnt = 0;
	for(uint32_t i = lo;i < hi;++i){
		const uint64_t k = fin_keys[i];
		if(k != ~0ull && (i == 0 || fin_keys[i - 1] != k)) ++cnt;
	}
	part[lane] = cnt;
	wave_sync();
	if(lane == 0){
		uint32_t run = 0;
		for(int t = 0;t < 64;++t){ const uint32_t c = part[t]; part[t] = run; run += c; }
		seg_hi[seq] = seq*cap + run;
	}
	wave_sync();
	uint32_t rank = part[lane];
	for(uint32_t i = lo;i < hi;++i){
		const uint64_t k = fin_keys[i];
		if(k != ~0ull && (i == 0 || fin_keys[i - 1] != k)){
			DevEntry e;
			materialise_entry(k, planes, blk_off, irr, irr_off, e);
			db[(size_t)seq*cap + rank] = e;
			++rank;
		}
	}
	wave_sync();                                 // the wave's LDS keys are reused by its next sequence
	}
}

// Buckets beyond MAX_BUCKET_CAP (low-complexity sequence against low-complexity oligos: thousands of tied
// windows): the same filter / sort / dedupe / materialise with the keys in global scratch and one
// 256-thread workgroup per touched sequence.  Slow path; exists so that such inputs are not refused.
constexpr int FINBIG_THREADS = 256;
constexpr uint32_t MAX_BUCKET_CAP_GLOBAL = 65536;
__global__ __launch_bounds__(FINBIG_THREADS) void k_finalize_big(const Hit *__restrict__ hits, const uint32_t *__restrict__ seq_count,
	uint32_t cap, const uint32_t *__restrict__ best, uint32_t ncand, const uint4 *__restrict__ planes,
	const uint64_t *__restrict__ blk_off, const IrrDev *__restrict__ irr, const uint32_t *__restrict__ irr_off,
	DevEntry *__restrict__ db, uint32_t *__restrict__ seg_hi, uint32_t *__restrict__ counters, uint32_t epoch,
	const uint32_t *__restrict__ touched, uint64_t *__restrict__ scratch)
{
	__shared__ uint32_t part[FINBIG_THREADS];
	if(blockIdx.x >= counters[3]) return;
	const uint32_t seq = touched[blockIdx.x], tid = threadIdx.x;
	uint32_t np2cap = 1; while(np2cap < cap) np2cap <<= 1;
	uint64_t *fin_keys = scratch + (size_t)blockIdx.x*np2cap;
	const uint32_t n = min(seq_count[seq], cap);
	uint32_t np2 = 1; while(np2 < n) np2 <<= 1;
	for(uint32_t i = tid;i < np2;i += FINBIG_THREADS){
		uint64_t k = ~0ull;
		if(i < n){
			const Hit h = hits[(size_t)seq*cap + i];
			if(((epoch << 8) | h.cnt) == best[(size_t)seq*ncand + h.cand]) k = h.key;
		}
		fin_keys[i] = k;
	}
	__syncthreads();
	for(uint32_t k = 2;k <= np2;k <<= 1){
		for(uint32_t j = k >> 1;j > 0;j >>= 1){
			for(uint32_t i = tid;i < np2;i += FINBIG_THREADS){
				const uint32_t l = i ^ j;
				if(l > i){
					const uint64_t a = fin_keys[i], b = fin_keys[l];
					const bool up = ((i & k) == 0);
					if((a > b) == up){ fin_keys[i] = b; fin_keys[l] = a; }
				}
			}
			__syncthreads();
		}
	}
	const uint32_t chunk = (np2 + FINBIG_THREADS - 1)/FINBIG_THREADS;
	const uint32_t lo = min(np2, tid*chunk), hi = min(np2, lo + chunk);
	uint32_t cnt = 0;
	for(uint32_t i = lo;i < hi;++i){
		const uint64_t k = fin_keys[i];
		if(k != ~0ull && (i == 0 || fin_keys[i - 1] != k)) ++cnt;
	}
	part[tid] = cnt;
	__syncthreads();
	if(tid == 0){
		uint32_t run = 0;
		for(int t = 0;t < FINBIG_THREADS;++t){ const uint32_t c = part[t]; part[t] = run; run += c; }
		seg_hi[seq] = seq*cap + run;
	}
	__syncthreads();
	uint32_t rank = part[tid];
	for(uint32_t i = lo;i < hi;++i){
		const uint64_t k = fin_keys[i];
		if(k != ~0ull && (i == 0 || fin_keys[i - 1] != k)){
			DevEntry e;
			materialise_entry(k, planes, blk_off, irr, irr_off, e);
			db[(size_t)seq*cap + rank] = e;
			++rank;
		}
	}
}

#include "pcr_screen.inc"

#include "pcr_sw.inc"
#include "pcr_thermo.inc"

// ============================================================================== host state
constexpr uint32_t N_TOUCHED_UNKNOWN = 0xFFFFFFFFu;
constexpr uint32_t EPOCH_LIMIT = (1u << 24) - 1;    // largest pass epoch that fits the 24-bit tag of best[]

template<class T> struct DevBuf {
	T *p = nullptr; size_t cap = 0;
	uint64_t generation = 0;     // bumped by every (re)allocation: the contents are undefined afterwards
	int ensure(size_t n)
	{
		if(n <= cap) return PCR_OK;
		++generation;
		if(p){ (void)hipFree(p); p = nullptr; cap = 0; }
		const size_t want = std::max<size_t>(n, 16);
		hipError_t e = hipMalloc((void **)&p, want*sizeof(T));
		if(e != hipSuccess){ g_err = std::string("hipMalloc: ") + hipGetErrorString(e); return PCR_ERR_DEVICE; }
		cap = want;
		return PCR_OK;
	}
	// for lists that grow a little at a time (the irregular words after every batch of splits): a quarter of slack, so that a hipFree +
	// hipMalloc pair -- a device-wide wait each -- is not paid on every growth
	int ensure_slack(size_t n) { return (n <= cap) ? PCR_OK : ensure(n + n/4 + 1024); }
	void release() { if(p){ (void)hipFree(p); p = nullptr; cap = 0; } }
};

struct SeqSet {
	uint32_t n = 0;
	std::vector<std::vector<uint8_t> > packed;   // host copy (split_sequence, irregular re-derivation)
	std::vector<uint64_t> len, blk_off, nblk_real;
	std::vector<float> weight; DevBuf<float> d_weight; bool weight_dirty = true;   // device copy on demand (pcr_optimize_batch)
	std::vector<uint8_t> active;
	std::vector<std::vector<pcrhost::IrrEntry> > irr_host;
	uint64_t total_blocks = 0;
	uint32_t n_tiles = 0, n_irr = 0;
	bool ctrl_clean = false;     // counters[0], counters[2], the per-sequence fills and the segment ends of sequences without hits are zero (the fused tail k_post leaves them so; counters[1] and [3] may be set by k_post / k_touched*: the lean k_seed2 zeroes those two itself): the next fused pass need not clear the block
	bool touched_from_seg = false;   // the per-sequence fills were zeroed by k_post: the touched list comes from the segment ends
	uint32_t bucket_cap = 64;   // hit slots per sequence of this set's word DB (grows on overflow; per set: the target DB at 0.9 needs 64, a background DB selected at 0.72 thousands)
	DevBuf<uint32_t> irr_perm; uint32_t irr_size_count[256];   // irregular words by size counter, largest first
	DevBuf<uint4> planes;
	DevBuf<uint32_t> valid, nib, tb, blk_seq, tile_seq, tile_pos0, irr_off, degen_tiles;
	// valid[] and tb[] carry zero padding in front and behind (k_seed2's tile fetch has no bounds checks, pcr_scan_seed2.inc): the DATA start here
	uint32_t *valid_d() const { return valid.p + S2_VALID_FRONT; }
	uint32_t *tb_d() const { return tb.p + S2_TB_FRONT; }
	DevBuf<uint8_t> tile_degen; uint32_t n_degen_tiles = 0;
	DevBuf<TileDesc> tile_desc;   // one per tile, with the sequence's active flag folded in (k_tile_desc; refreshed by pcr_set_active)
	DevBuf<uint64_t> d_len, d_blk_off, d_nblk_real;
	DevBuf<uint8_t> d_active;
	std::vector<uint8_t> has_eos; DevBuf<uint8_t> d_has_eos;   // sequence holds an EOS nibble (record padding, splits): only then has_split has to look
	DevBuf<IrrDev> irr;
	DevBuf<IrrScan> irr_scan;     // the irregular words in scan order with their 2-bit codes (k_seed2)
	DevBuf<uint32_t> irx_first, irx_last, irx_words, irx_sums; bool irx_valid = false, irx_usable = false;   // their inverse index (pcr_scan_seed2.inc), built on demand; usable: no key's run is longer than IRX_MAX_RUN
	uint32_t irr_n_multi = 0;     // irregular words holding an IUPAC slot (they meet every candidate: no index for them)
	// the positions of the set by the 9-gram that starts there (pcr_scan_seed3.inc), built on demand after a load
	DevBuf<uint32_t> pix_first, pix_last, pix_sums, blk_info, blk_local, blk_tile0; DevBuf<uint4> pix_ent; std::vector<uint32_t> pix_count_h; uint64_t pix_generation = 0; bool pix_valid = false, pix_usable = false;
	DevBuf<uint8_t> codes; DevBuf<uint64_t> d_code_off; bool have_codes = false;
	// word DB of the last select
	bool have_db = false;
	uint32_t n_entries = 0;     // DB entries over all sequences
	uint32_t db_cap = 0;        // slots per sequence in `db`; the entries of sequence s are db[s*db_cap .. seg_hi[s])
	uint64_t n_slots = 0;       // n * db_cap
	DevBuf<DevEntry> db;
	DevBuf<uint32_t> touched;     // sequences holding DB entries (k_touched)
	DevBuf<uint32_t> ctrl;        // [0..7] counters | [8, 8+n) per-sequence hit counts | [8+n, 8+2n) seg_hi : one memset per pass
	uint32_t n_touched = 0;
	bool touched_built = false;   // touched[] / counters[3] hold the list of the last pass (the fused tail does not build it)
	uint32_t *d_seg_hi = nullptr;
	void release()
	{
		d_weight.release(); tile_desc.release(); irr_scan.release(); irx_first.release(); irx_last.release(); irx_words.release(); irx_sums.release(); pix_first.release(); pix_last.release(); pix_ent.release(); pix_sums.release(); blk_info.release(); blk_local.release(); blk_tile0.release(); irr_perm.release(); planes.release(); valid.release(); nib.release(); tb.release(); degen_tiles.release(); tile_degen.release(); blk_seq.release(); tile_seq.release(); tile_pos0.release();
		irr_off.release(); d_len.release(); d_blk_off.release();
		d_nblk_real.release(); d_active.release(); d_has_eos.release(); irr.release(); db.release(); touched.release(); ctrl.release(); codes.release(); d_code_off.release();
	}
};

} // namespace

// A few host threads that live as long as the handle (pcr_optimize.inc's loops over the assays of a batch).  run(): part indices
// 0 .. n_parts-1 are claimed one by one by the workers and by the calling thread, which returns when all are done.
struct HostPool {
	std::vector<std::thread> th;
	std::mutex m; std::condition_variable cv_go, cv_done;
	std::function<void(unsigned)> fn; unsigned n_parts = 0, next = 0, pending = 0; uint64_t gen = 0; bool stop = false;
	explicit HostPool(unsigned n_workers){ for(unsigned i = 0;i < n_workers;++i) th.emplace_back([this]{ work(); }); }
	~HostPool(){ { std::lock_guard<std::mutex> lk(m); stop = true; } cv_go.notify_all(); for(std::thread &t : th) t.join(); }
	unsigned size() const { return (unsigned)th.size(); }
	void work()
	{
		uint64_t seen = 0;
		std::unique_lock<std::mutex> lk(m);
		while(true){
			cv_go.wait(lk, [&]{ return stop || gen != seen; });
			if(stop) return;
			seen = gen;
			while(next < n_parts){ const unsigned p = next++; lk.unlock(); fn(p); lk.lock(); }
			if(--pending == 0) cv_done.notify_one();
		}
	}
	void run(unsigned parts, std::function<void(unsigned)> f)
	{
		std::unique_lock<std::mutex> lk(m);
		fn = std::move(f); n_parts = parts; next = 0; pending = (unsigned)th.size(); ++gen;
		cv_go.notify_all();
		while(next < n_parts){ const unsigned p = next++; lk.unlock(); fn(p); lk.lock(); }
		cv_done.wait(lk, [&]{ return pending == 0; });
	}
};

struct pcr_ctx {
	int device = 0;
	std::unique_ptr<HostPool> pool_;
	HostPool &host_pool()
	{
		if(!pool_){ const unsigned hw = std::max(1u, std::thread::hardware_concurrency()); pool_.reset(new HostPool(std::min(hw, 16u) - 1u)); }   // (a GPU box gives its process 16 cores)
		return *pool_;
	}
	hipStream_t stream = nullptr;
	bool own_stream = false;
	std::string design_text;            // the output file of the last pcr_design call
	DevBuf<uint64_t> split_where;       // pcr_split_many's (block, bit) list
	hipStream_t aux_stream = nullptr;   // the optimiser's thermodynamics run here, beside the coverage passes on `stream` (pcr_optimize.inc); created on first use
	pcr_params params;
	pcrhost::PackFilter filt;
	SeqSet sets[PCR_N_SETS];   // target, background, multiplex (accepted amplicons), scratch (pcr_multiplex_screen's trial amplicons)
	// scratch
	DevBuf<uint32_t> best, counters, mask, status;
	int scan_version = 3;
	bool force_seed1 = false;   // PCRAMP_SEED=1: the first form of the seed scan (A/B)
	// second form of the seed scan: the pass's seed list, mask tables (reused between passes) and the per-oligo seed cache
	std::vector<uint32_t> s2_seeds; std::vector<uint4> s2_masks; std::vector<uint8_t> s2_floors;
	std::vector<uint32_t> s2_group_end, s2_group_offmask, s2_group_or, s2_group_nor;   // per group: end in the seed list, forward-seed slot offsets, first orientation, orientations spanned   // the seed list in groups of whole orientations, each within one launch's LDS budget
	struct S2Key { uint32_t a, c, g, t, floor_; bool operator==(const S2Key &o) const { return a == o.a && c == o.c && g == o.g && t == o.t && floor_ == o.floor_; } };
	struct S2KeyHash { size_t operator()(const S2Key &k) const { uint64_t h = 0x9E3779B97F4A7C15ull; for(uint32_t v : {k.a, k.c, k.g, k.t, k.floor_}){ h ^= v; h *= 0x100000001B3ull; h ^= h >> 29; } return (size_t)h; } };
	struct S2Entry { std::vector<uint32_t> seeds; /* code << 14 | off << 9 */ uint32_t off_mask; bool seedable; uint4 mask; /* the orientation's interleaved mask entry (k_seed2 / k_seed3) */
		std::vector<uint32_t> chunks; uint32_t chunks_total = 0; uint64_t chunks_gen = 0; /* third form: running 64-entry chunks of the seeds' runs in the set whose position index has this generation */ };
	std::unordered_map<S2Key, S2Entry, S2KeyHash> s2_cache;
	std::vector<pcrhost::Seed> s2_tmp;
	struct S3Launch { Seed3Slices W; uint32_t n_chunks, per_wg, slice_cap, n_irr_wg; };
	std::vector<S3Launch> s3_launch;               // third form: per launch group, the workgroups' slices of the chunk list
	std::vector<uint32_t> s3_prefix; bool no_seed3 = false, s3_attr_set = false;   // third form: the pass's chunk list; PCRAMP_SEED3=0: second form (A/B)
	bool s2_attr_set = false; uint32_t s2_dbg = 0; bool no_irr_index = false;   // PCRAMP_IRR_INDEX=0: the irregular words scanned in chunks by every wave (A/B)
	// first form, tables built on the device (k_seed_tables): the pass's seed list, its own per-oligo cache (8-gram seeds), the tables
	std::vector<uint32_t> s1_seeds; std::unordered_map<S2Key, S2Entry, S2KeyHash> s1_cache;
	DevBuf<uint32_t> s1_image, s1_heads, s1_multi, s1_part;
	bool host_seed_tables = false;   // PCRAMP_SEED_TABLES=host: build them on the host as for passes with shift candidates (A/B)
	DevBuf<Hit> hits;
	DevBuf<uint64_t> bits_fr, bits_rf;
	DevBuf<OligoDev> oligos;
	DevBuf<SwJob> sw_jobs; DevBuf<SwOut> sw_out; DevBuf<uint8_t> sw_q, sw_qlen, sw_t, entry_codes, entry_lens;
	DevBuf<AmpRec> amp_recs, amp_recs2; DevBuf<BgPairDev> bg_pairs;
	DevBuf<uint64_t> amp_keys; DevBuf<uint32_t> amp_pkeys, amp_pair_start; DevBuf<uint8_t> sort_tmp;   // reference-order sort of the candidate amplicons (order_amplicons)
	DevBuf<thermo::Job> th_jobs; DevBuf<thermo::JobOut> th_out; DevBuf<int> th_dg; DevBuf<uint2> th_map; DevBuf<uint32_t> th_bad;
	float th_dg_salt = -1.0f;   // salt the table in th_dg was built for
	DevBuf<unsigned long long> th_dbg; bool th_attr_set = false;
	std::vector<thermo::Job> th_host_jobs; std::vector<thermo::JobOut> th_host_res;   // pcr_thermo's job records and results, kept between calls
	uint8_t *sw_pin = nullptr, *sw_pin_dev = nullptr; hipEvent_t sw_done[2] = {nullptr, nullptr};   // pcr_sw_align_words: two pinned chunk buffers (words in, results out)
	DevBuf<pcr_amplicon> mx_amp;   // pcr_collect_amplicons records
	DevBuf<OligoDev> opt_oligos; DevBuf<uint2> opt_jobs; DevBuf<float> opt_cov; DevBuf<uint32_t> opt_loc, opt_tasks;   // pcr_optimize_batch: base oligos + trial words, per-oligo variant ranges, coverages
	DevBuf<Planes> mx_keys; uint32_t mx_n_keys = 0; DevBuf<uint32_t> mx_count;   // multiplex background: unique words of the accepted amplicons (pcr_multiplex.inc)
	size_t amp_cap = size_t(1) << 20;
	uint32_t n_cu = 256;        // compute units of the device (hipDeviceProp)
	DevBuf<uint64_t> fin_scratch;   // k_finalize_big's keys
	std::vector<uint16_t> seed_count; std::vector<uint8_t> seed_fill, seed_own;   // host scratch of the seed-table builder
	uint32_t epoch = 0;         // pass counter tagging best[] (see HitSink)
	uint32_t debug_epoch = 0;   // PCRAMP_DEBUG_EPOCH: value the counter takes when best[] is first cleared
	bool opt_pm_global = false;   // PCRAMP_OPT_PM=global: k_pair_moves_batch (partner walk through global memory) instead of k_pair_moves_lds (A/B)
	uint32_t opt_dbg_wg_tasks = 0, opt_dbg_task_cap = 0;   // PCRAMP_DEBUG_OPT_TASKS=<per workgroup>,<global>: shrink k_pair_moves_batch's task lists so that their overflow branches run (tests)
	uint64_t best_seen = 0;      // generation of best[] that has been cleared (the allocator may hand the same address back: never compare pointers)
	// pinned staging for the small per-call host->device payload (candidates, tables, oligos): one async copy
	// ring of host-mapped staging buffers: a slot is rewritten only after the k_stage that read it has run,
	// so the host can prepare the next pass while the previous one is still on the GPU
	static constexpr int STAGE_RING = 4;
	struct StageSlot { uint8_t *host = nullptr, *dev = nullptr; size_t cap = 0; hipEvent_t done = nullptr; bool busy = false;
	                   uint32_t guard_seq = 0; /* != 0: free once the pass with this mailbox sequence number has published */ };
	StageSlot stage[STAGE_RING]; int stage_next = 0;
	// The lean form of the fused pass needs no staging launch at all: its tables are written by the CPU straight into
	// fine-grained DEVICE memory (large BAR: posted writes, ordered before the doorbell of the launch that follows; measured
	// 0.6 us for 32 KB, profiles/microbench/bar_write.hip), a ring of its own because the kernels read the slot itself.
	StageSlot dstage[STAGE_RING]; int dstage_next = 0;
	bool direct_ok = false;       // fine-grained device memory can be had and written (probed in pcr_create; PCRAMP_STAGE=kernel turns it off)
	typedef PassMail Mail;
	static constexpr uint32_t MAIL_RING = 8;
	Mail *mail = nullptr, *mail_dev = nullptr;   // host-mapped ring (slot = seq % MAIL_RING): k_publish writes it, the host spins on seq
	uint32_t mail_seq = 0;
	// results of the small synchronous calls (move coverage, thermodynamics) come back the same way: a kernel
	// copies them into a host-mapped buffer and raises a flag the host spins on (no copy-engine packets, no
	// interrupt wake-up; three pageable hipMemcpyAsync + a stream sync cost ~1.8 ms per call instead)
	uint8_t *ret_host = nullptr, *ret_dev = nullptr; size_t ret_cap = 0;
	uint8_t *in_host = nullptr, *in_dev = nullptr; size_t in_cap = 0;   // the way in for small synchronous calls: a mapped buffer the kernel reads directly (the call waits for its results, so one buffer is enough)
	uint32_t *ret_flag = nullptr, *ret_flag_dev = nullptr; uint32_t ret_seq = 0;
	// passes enqueued by pcr_screen_device whose counters have not been looked at yet (pcr_synchronize / any
	// other entry point drains them; a bucket overflow found then replays the passes synchronously)
	struct Pending {
		uint32_t seq; int which; std::vector<pcr_pair> pairs; int opt5, opt3; float thr; uint32_t min_len;
		pcr_amplify_args args; uint64_t *d_fr, *d_rf;
	};
	std::vector<Pending> pending;
	DevBuf<uint8_t> arena;
	const uint4 *d_cand_fwd = nullptr, *d_cand_rc = nullptr; const uint32_t *d_cand_floor = nullptr;
	const OligoDev *d_oligos = nullptr;
	// host-side phase timers (PCRAMP_TIMING=1: printed by pcr_destroy)
	bool debug_log = false;                       // PCRAMP_DEBUG: plan / launch lines on stderr (read once: getenv walks the environment)
	bool timing = false; double t_host[8] = {0, 0, 0, 0, 0, 0, 0, 0}; uint64_t n_timed = 0;
	// profiling
	bool prof = false; uint32_t prof_stride = 1, prof_pass = 0;   // events bracket the scan of every prof_stride-th pass
	std::vector<std::pair<hipEvent_t, hipEvent_t> > prof_events;
	double prof_ms = 0.0; uint64_t prof_launches = 0;
	// the same for the other kernels bench.py prices (PCR_PROF_SW, PCR_PROF_THERMO): every launch while profiling is on
	std::vector<std::pair<hipEvent_t, hipEvent_t> > prof_events_k[PCR_PROF_KERNELS];
	double prof_ms_k[PCR_PROF_KERNELS] = {0.0, 0.0, 0.0}; uint64_t prof_launches_k[PCR_PROF_KERNELS] = {0, 0, 0};
};

namespace {

int drain(pcr_ctx *ctx);

// HIP events around the launches of a priced kernel (PCR_PROF_*), on the launch stream, while profiling is on.  The pair is
// handed to the context when the scope ends (also on an error return), so no event is ever left behind.
struct ProfScope {
	pcr_ctx *ctx; int k; hipEvent_t e0 = nullptr, e1 = nullptr;
	ProfScope(pcr_ctx *c, int kernel, bool on = true) : ctx(c), k(kernel)
	{
		if(!on || !ctx->prof || k < 0 || k >= PCR_PROF_KERNELS) return;
		if(hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess){ if(e0) (void)hipEventDestroy(e0); e0 = e1 = nullptr; return; }
		(void)hipEventRecord(e0, ctx->stream);
	}
	void finish()
	{
		if(!e0) return;
		(void)hipEventRecord(e1, ctx->stream);
		((k == PCR_PROF_SCAN) ? ctx->prof_events : ctx->prof_events_k[k]).push_back(std::make_pair(e0, e1));
		e0 = e1 = nullptr;
	}
	~ProfScope() { finish(); }
};
// the sets a caller may name (the fourth slot is internal)
constexpr int PCR_SET_SCRATCH = 3;
inline bool set_ok(pcr_set which) { return (unsigned)which <= (unsigned)PCR_SET_MULTIPLEX; }
#define CHECK_SET(which) do{ if(!set_ok(which)){ g_err = "unknown sequence set (PCR_SET_TARGET, PCR_SET_BACKGROUND or PCR_SET_MULTIPLEX)"; return PCR_ERR_ARG; } }while(0)
#define DRAIN(ctx) do{ if(!(ctx)->pending.empty()){ const int drc_ = drain(ctx); if(drc_ != PCR_OK) return drc_; } }while(0)

struct HostTimer {
	pcr_ctx *ctx; int slot; std::chrono::steady_clock::time_point t0;
	HostTimer(pcr_ctx *c, int s) : ctx(c), slot(s) { if(ctx->timing) t0 = std::chrono::steady_clock::now(); }
	void next(int s)
	{
		if(!ctx->timing) return;
		const auto t1 = std::chrono::steady_clock::now();
		ctx->t_host[slot] += std::chrono::duration<double, std::micro>(t1 - t0).count();
		slot = s; t0 = t1;
	}
	~HostTimer() { next(slot); }
};

// pcr_split_many: where[2i] = block of the plane store, where[2i+1] = bit of the base in it.  Several splits may share a block.
__global__ void k_apply_splits(const uint64_t *__restrict__ where, uint32_t n, uint32_t *__restrict__ planes /* uint4 per block */, uint32_t *__restrict__ nib)
{
	const uint32_t i = blockIdx.x*blockDim.x + threadIdx.x;
	if(i >= n) return;
	const uint64_t gb = where[2*(size_t)i];
	const uint32_t bit = (uint32_t)where[2*(size_t)i + 1];
	const uint32_t m = ~(1u << bit);
	for(int k = 0;k < 4;++k) atomicAnd(&planes[gb*4 + k], m);
	atomicAnd(&nib[gb*4 + (bit >> 3)], ~(0xFu << ((bit & 7u)*4u)));
}

int upload_irregular(pcr_ctx *ctx, SeqSet &S)
{
	const auto tu0 = std::chrono::steady_clock::now();
	auto ms_since = [&](){ return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tu0).count(); };
	double t_flat = 0, t_sort = 0, t_scan = 0, t_copy1 = 0;
	size_t total = 0;
	for(uint32_t s = 0;s < S.n;++s) total += S.irr_host[s].size();
	std::vector<IrrDev> flat(total);
	std::vector<uint32_t> off(S.n + 1, 0);
	for(int k = 0;k < 256;++k) S.irr_size_count[k] = 0;
	{
		size_t at = 0;
		for(uint32_t s = 0;s < S.n;++s){
			off[s] = (uint32_t)at;
			const std::vector<pcrhost::IrrEntry> &v = S.irr_host[s];
			for(size_t k = 0;k < v.size();++k){
				const pcrhost::IrrEntry &e = v[k];
				IrrDev &d = flat[at++];
				d.w = e.w; d.loc = e.loc; d.seq = s;
				d.meta = (uint32_t)e.strand | ((uint32_t)e.cws << 8) | ((uint32_t)e.ord << 16);
				d.local_id = (uint32_t)k;
				++S.irr_size_count[e.cws & 0xFF];
			}
		}
	}
	off[S.n] = (uint32_t)flat.size();
	S.n_irr = (uint32_t)flat.size();
	t_flat = ms_since();
	// scan order: by size counter, largest first, so that a pass with min_oligo_length m walks a prefix (sequence.cpp:157,239 drop
	// the words whose counter is below m); equal counters keep the list's order.  A counting sort: the list is rebuilt after every
	// accepted assay of a design run (pcr_split_many), and a comparison sort of C2's 1.2e6 words was most of the 280 ms that took.
	std::vector<uint32_t> perm(flat.size());
	t_sort = ms_since();
	int rc;
	{
		// bits of a 32-bit plane to the even positions of a 64-bit word
		auto spread = [](uint64_t v) -> uint64_t {
			v = (v | (v << 16)) & 0x0000FFFF0000FFFFull; v = (v | (v << 8)) & 0x00FF00FF00FF00FFull; v = (v | (v << 4)) & 0x0F0F0F0F0F0F0F0Full;
			v = (v | (v << 2)) & 0x3333333333333333ull; v = (v | (v << 1)) & 0x5555555555555555ull; return v;
		};
		// one pass in list order: a word goes to the next free place of its size counter's run, and its scan record is written there
		// (reads in sequence, writes into a handful of runs that each grow in sequence; gathering the records in scan order
		// instead -- a random 48-byte read per word -- was 48 of the 61 ms this took for C2's 1.2e6 words)
		uint32_t start[256];
		uint32_t run = 0;
		for(int k = 255;k >= 0;--k){ start[k] = run; run += S.irr_size_count[k]; }
		std::vector<IrrScan> scan(perm.size());
		uint32_t n_multi = 0;
		for(size_t i = 0;i < flat.size();++i){
			const IrrDev &d = flat[i];
			const uint32_t at = start[(d.meta >> 8) & 0xFF]++;
			perm[at] = (uint32_t)i;
			const uint32_t multi = (d.w.a & d.w.c) | (d.w.a & d.w.g) | (d.w.a & d.w.t) | (d.w.c & d.w.g) | (d.w.c & d.w.t) | (d.w.g & d.w.t);
			if(multi) ++n_multi;
			const uint32_t lo = d.w.c | d.w.t, hi = d.w.g | d.w.t;             // 2-bit code planes (A,C,G,T = 0..3; empty slots read as A)
			const uint64_t code = spread(lo) | (spread(hi) << 1);
			IrrScan r; r.w0 = (uint32_t)code; r.w1 = (uint32_t)(code >> 32); r.idx_flags = (uint32_t)i | (multi ? 0x80000000u : 0u); r.seq = d.seq;
			scan[at] = r;
		}
		S.irr_n_multi = n_multi; S.irx_valid = false;
		t_scan = ms_since();
		if((rc = S.irr_scan.ensure_slack(scan.size() + 1)) != PCR_OK) return rc;
		if(!scan.empty()) HIP_TRY(hipMemcpy(S.irr_scan.p, scan.data(), scan.size()*sizeof(IrrScan), hipMemcpyHostToDevice));
	}
	t_copy1 = ms_since();
	if((rc = S.irr_perm.ensure_slack(perm.size() + 1)) != PCR_OK) return rc;
	if(!perm.empty()) HIP_TRY(hipMemcpyAsync(S.irr_perm.p, perm.data(), perm.size()*sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
	if((rc = S.irr.ensure_slack(flat.size())) != PCR_OK) return rc;
	if((rc = S.irr_off.ensure(off.size())) != PCR_OK) return rc;
	if(!flat.empty()) HIP_TRY(hipMemcpyAsync(S.irr.p, flat.data(), flat.size()*sizeof(IrrDev), hipMemcpyHostToDevice, ctx->stream));
	HIP_TRY(hipMemcpyAsync(S.irr_off.p, off.data(), off.size()*sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
	HIP_TRY(hipStreamSynchronize(ctx->stream));
	if(ctx->timing && flat.size() > 100000) fprintf(stderr, "[pcramp] irregular list of %zu words: flat %.1f  order %.1f  scan records %.1f  first copy %.1f  other copies %.1f ms\n", flat.size(), t_flat, t_sort - t_flat, t_scan - t_sort, t_copy1 - t_scan, ms_since() - t_copy1);
	return PCR_OK;
}

// The inverse index of the set's irregular words (pcr_scan_seed2.inc), for the state the set is in: counting sort of (word, slot
// offset) by the 9-gram read there.  ~1 ms for C2's 1.2e6 words; rebuilt after the irregular list changes (load, splits).
constexpr uint32_t IRX_MAX_RUN = 1024;
int ensure_irr_index(pcr_ctx *ctx, SeqSet &S)
{
	if(S.irx_valid) return PCR_OK;
	int rc;
	if((rc = S.irx_first.ensure(IRX_KEYS + 4)) != PCR_OK) return rc;
	if((rc = S.irx_last.ensure(IRX_KEYS + 4)) != PCR_OK) return rc;
	if((rc = S.irx_words.ensure_slack((size_t)24*S.n_irr + 4)) != PCR_OK) return rc;
	const uint32_t n_blocks = (IRX_KEYS + 4095u)/4096u;                           // 1 536
	if((rc = S.irx_sums.ensure(n_blocks + 4)) != PCR_OK) return rc;
	HIP_TRY(hipMemsetAsync(S.irx_last.p, 0, (size_t)IRX_KEYS*sizeof(uint32_t), ctx->stream));   // counts first, ends of the runs in the end
	if(S.n_irr){
		hipLaunchKernelGGL(k_irx_count, dim3((S.n_irr + 255)/256), dim3(256), 0, ctx->stream, S.irr.p, S.n_irr, S.irx_last.p);
		HIP_TRY(hipGetLastError());
	}
	{
		// a key shared by more words than a wave should walk (IRX_MAX_RUN): the set keeps the chunk form of the irregular scan
		uint32_t longest = 0;
		HIP_TRY(hipMemsetAsync(S.irx_sums.p + n_blocks, 0, sizeof(uint32_t), ctx->stream));
		hipLaunchKernelGGL(k_irx_max, dim3(256), dim3(256), 0, ctx->stream, S.irx_last.p, IRX_KEYS, S.irx_sums.p + n_blocks);
		HIP_TRY(hipMemcpyAsync(&longest, S.irx_sums.p + n_blocks, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
		HIP_TRY(hipStreamSynchronize(ctx->stream));
		S.irx_usable = longest <= IRX_MAX_RUN;
		if(!S.irx_usable){ S.irx_valid = true; return PCR_OK; }
	}
	hipLaunchKernelGGL(k_scan_blocks, dim3(n_blocks), dim3(1024), 0, ctx->stream, S.irx_last.p, S.irx_first.p, IRX_KEYS, S.irx_sums.p);
	hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, ctx->stream, S.irx_sums.p, n_blocks);
	hipLaunchKernelGGL(k_scan_add, dim3(n_blocks), dim3(1024), 0, ctx->stream, S.irx_first.p, S.irx_last.p, IRX_KEYS, S.irx_sums.p);   // first = starts; last = a copy, advanced by the scatter to the ends
	HIP_TRY(hipGetLastError());
	if(S.n_irr){
		hipLaunchKernelGGL(k_irx_scatter, dim3((S.n_irr + 255)/256), dim3(256), 0, ctx->stream, S.irr.p, S.n_irr, S.irx_last.p, S.irx_words.p);
		HIP_TRY(hipGetLastError());
	}
	S.irx_valid = true;
	return PCR_OK;
}

// The set's positions by the 9-gram that starts there (pcr_scan_seed3.inc), each with the 64 bases around it: counting sort on the
// device.  20 bytes per base; built once per load.
int ensure_pos_index(pcr_ctx *ctx, SeqSet &S)
{
	if(S.pix_valid) return PCR_OK;
	S.pix_valid = true; S.pix_usable = false;
	if(S.total_blocks == 0 || S.total_blocks*32 >= (uint64_t(1) << 32) - 64) return PCR_OK;   // positions are 32-bit
	int rc;
	if((rc = S.pix_first.ensure(PIX_CODES + 4)) != PCR_OK) return rc;
	if((rc = S.pix_last.ensure(PIX_CODES + 4)) != PCR_OK) return rc;
	const uint32_t n_blocks = (PIX_CODES + 4095u)/4096u;
	if((rc = S.pix_sums.ensure(n_blocks + 4)) != PCR_OK) return rc;
	if((rc = S.pix_ent.ensure(S.total_blocks*32 + 64)) != PCR_OK) return rc;
	{
		// per 32-base block: its sequence (bit 31: the tile it lies in holds IUPAC codes) and its number within the sequence -- what a
		// window that reached its floor is asked, in one round trip
		if(S.n >= (1u << 30)){ return PCR_OK; }
		if((rc = S.blk_info.ensure(S.total_blocks + 1)) != PCR_OK) return rc;
		if((rc = S.blk_local.ensure(S.total_blocks + 1)) != PCR_OK) return rc;
		// (made on the device: a C4 shard has 10^8 blocks)
		std::vector<uint32_t> tile0((size_t)S.n + 1);
		uint64_t t0 = 0;
		for(uint32_t q = 0;q < S.n;++q){ tile0[q] = (uint32_t)t0; t0 += (S.len[q] >= 32) ? (S.len[q] - 7 + TILE_POS - 1)/TILE_POS : 0; }
		tile0[S.n] = (uint32_t)t0;
		if(t0 >= (uint64_t(1) << 32)) return PCR_OK;
		if((rc = S.blk_tile0.ensure((size_t)S.n + 1)) != PCR_OK) return rc;
		HIP_TRY(hipMemcpyAsync(S.blk_tile0.p, tile0.data(), ((size_t)S.n + 1)*sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
		hipLaunchKernelGGL(k_blk_info, dim3((unsigned)((S.total_blocks + 255)/256)), dim3(256), 0, ctx->stream, S.blk_seq.p, S.d_blk_off.p, S.d_len.p,
			S.blk_tile0.p, S.tile_degen.p, S.d_active.p, S.total_blocks, S.blk_info.p, S.blk_local.p);
		HIP_TRY(hipGetLastError());
		HIP_TRY(hipStreamSynchronize(ctx->stream));                                  // (tile0 is a local)
	}
	HIP_TRY(hipMemsetAsync(S.pix_last.p, 0, (size_t)PIX_CODES*sizeof(uint32_t), ctx->stream));
	const unsigned grid = (unsigned)((S.total_blocks + 255)/256);
	hipLaunchKernelGGL(k_pix_build<false>, dim3(grid), dim3(256), 0, ctx->stream, S.tb_d(), S.blk_seq.p, S.d_blk_off.p, S.d_len.p, S.total_blocks, S.pix_last.p, (uint4 *)nullptr);
	hipLaunchKernelGGL(k_scan_blocks, dim3(n_blocks), dim3(1024), 0, ctx->stream, S.pix_last.p, S.pix_first.p, PIX_CODES, S.pix_sums.p);
	hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, ctx->stream, S.pix_sums.p, n_blocks);
	hipLaunchKernelGGL(k_scan_add, dim3(n_blocks), dim3(1024), 0, ctx->stream, S.pix_first.p, S.pix_last.p, PIX_CODES, S.pix_sums.p);
	hipLaunchKernelGGL(k_pix_build<true>, dim3(grid), dim3(256), 0, ctx->stream, S.tb_d(), S.blk_seq.p, S.d_blk_off.p, S.d_len.p, S.total_blocks, S.pix_last.p, S.pix_ent.p);
	HIP_TRY(hipGetLastError());
	{
		std::vector<uint32_t> first(PIX_CODES), last(PIX_CODES);
		HIP_TRY(hipMemcpyAsync(first.data(), S.pix_first.p, (size_t)PIX_CODES*sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
		HIP_TRY(hipMemcpyAsync(last.data(), S.pix_last.p, (size_t)PIX_CODES*sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
		HIP_TRY(hipStreamSynchronize(ctx->stream));
		S.pix_count_h.resize(PIX_CODES);
		for(uint32_t c = 0;c < PIX_CODES;++c) S.pix_count_h[c] = last[c] - first[c];
		static uint64_t generations = 0;
		S.pix_generation = ++generations;
	}
	S.pix_usable = true;
	return PCR_OK;
}

int run_valid(pcr_ctx *ctx, SeqSet &S, uint64_t first_block, uint64_t n_blocks)
{
	if(n_blocks == 0) return PCR_OK;
	const unsigned threads = 256;
	const unsigned grid = (unsigned)((n_blocks + threads - 1)/threads);
	hipLaunchKernelGGL(k_valid, dim3(grid), dim3(threads), 0, ctx->stream, S.planes.p, S.d_blk_off.p, S.blk_seq.p,
		S.d_nblk_real.p, S.valid_d(), S.total_blocks, ctx->filt.max_degen, ctx->filt.gc_ok, first_block, n_blocks);
	HIP_TRY(hipGetLastError());
	return PCR_OK;
}

int build_tile_desc(pcr_ctx *ctx, SeqSet &S)
{
	if(S.n_tiles == 0) return PCR_OK;
	int rc = S.tile_desc.ensure(S.n_tiles);
	if(rc != PCR_OK) return rc;
	hipLaunchKernelGGL(k_tile_desc, dim3((S.n_tiles + 255)/256), dim3(256), 0, ctx->stream, S.d_blk_off.p, S.d_len.p, S.d_active.p, S.tile_seq.p,
		S.tile_pos0.p, S.tile_degen.p, S.n_tiles, S.tile_desc.p);
	HIP_TRY(hipGetLastError());
	return PCR_OK;
}

void fill_oligo_planes(OligoDev &o, const Planes &m, float thr2)
{
	o.m = m;
	const unsigned size = (unsigned)pcrhost::planes_size(o.m);
	o.floor2 = (unsigned)((float)size*thr2);                                     // optimize.cpp:293
	o.norm = (size > 0) ? (float)(1.0/size) : 0.0f;                              // optimize.cpp:221
	o.start = pcrhost::planes_start(o.m);
	o.stop = pcrhost::planes_stop(o.m);
	o.p1 = (o.stop >= 1) ? pcrhost::planes_nibble(o.m, o.stop - 1) : 0;
	o.p2 = (o.stop >= 0) ? pcrhost::planes_nibble(o.m, o.stop) : 0;
}

void fill_oligo(OligoDev &o, const uint64_t w[2], float thr2)
{
	o.m = pcrhost::planes_of_word(w);
	const unsigned size = (unsigned)pcrhost::planes_size(o.m);
	o.floor2 = (unsigned)((float)size*thr2);                                     // optimize.cpp:293
	o.norm = (size > 0) ? (float)(1.0/size) : 0.0f;                              // optimize.cpp:221
	o.start = pcrhost::planes_start(o.m);
	o.stop = pcrhost::planes_stop(o.m);
	o.p1 = (o.stop >= 1) ? pcrhost::planes_nibble(o.m, o.stop - 1) : 0;
	o.p2 = (o.stop >= 0) ? pcrhost::planes_nibble(o.m, o.stop) : 0;
}

// Packs small host arrays into a host-mapped pinned buffer; ONE kernel (k_stage) pulls them into the
// device arena and clears up to two device regions in the same launch -- a copy-engine transfer and a
// fill each cost ~4 us plus a ~6 us queue bubble on either side, a kernel runs back to back with the next.
__global__ void k_stage(const uint4 *__restrict__ src, uint4 *__restrict__ dst, uint32_t n16,
	uint4 *__restrict__ z0, uint32_t n0, uint4 *__restrict__ z1, uint32_t n1, uint4 *__restrict__ z2, uint32_t n2)
{
	const uint32_t i = blockIdx.x*blockDim.x + threadIdx.x, stride = gridDim.x*blockDim.x;
	for(uint32_t k = i;k < n16;k += stride) dst[k] = src[k];
	const uint4 zero = make_uint4(0, 0, 0, 0);
	for(uint32_t k = i;k < n0;k += stride) z0[k] = zero;
	for(uint32_t k = i;k < n1;k += stride) z1[k] = zero;
	for(uint32_t k = i;k < n2;k += stride) z2[k] = zero;
}

// Has the pass with mailbox sequence number `seq` published?  Publications happen in stream order and slot
// seq % MAIL_RING is reused by seq + MAIL_RING, ...: any value of that slot at or beyond seq means yes.
int wait_published(pcr_ctx *ctx, uint32_t seq)
{
	const PassMail *const slot = ctx->mail + (seq % pcr_ctx::MAIL_RING);
	uint64_t spins = 0;
	while(true){
		const uint32_t v = __atomic_load_n((const uint32_t *)&slot->seq, __ATOMIC_ACQUIRE);
		if(v >= seq && (v - seq) % pcr_ctx::MAIL_RING == 0) return PCR_OK;
		__builtin_ia32_pause();
		if((++spins & 0x3FFF) == 0){
			const hipError_t e = hipStreamQuery(ctx->stream);
			if(e == hipSuccess) return PCR_OK;                       // stream idle: whatever used the slot is over (the pass may have failed before publishing)
			if(e != hipErrorNotReady){ g_err = std::string("device pass failed: ") + hipGetErrorString(e); return PCR_ERR_DEVICE; }
		}
	}
}

struct Stager {
	pcr_ctx *ctx; size_t used = 0; bool direct = false;
	explicit Stager(pcr_ctx *c) : ctx(c) {}
	pcr_ctx::StageSlot *slot = nullptr;
	// direct: the bytes go straight into a slot of device memory (see pcr_ctx::dstage); the caller launches no k_stage for
	// them and passes the sequence number of ITS pass to seal(): the slot is free again once the NEXT pass has published,
	// i.e. once every kernel of this one is over.
	int begin_direct(size_t bytes)
	{
		direct = true;
		slot = &ctx->dstage[ctx->dstage_next];
		ctx->dstage_next = (ctx->dstage_next + 1) % pcr_ctx::STAGE_RING;
		if(slot->guard_seq){ const int grc = wait_published(ctx, slot->guard_seq); if(grc != PCR_OK) return grc; slot->guard_seq = 0; }
		if(bytes + 256 > slot->cap){
			if(slot->dev){ HIP_TRY(hipStreamSynchronize(ctx->stream)); (void)hipFree(slot->dev); slot->dev = slot->host = nullptr; slot->cap = 0; }
			const size_t want = std::max<size_t>((bytes + 256)*2, 1 << 17);
			HIP_TRY(hipExtMallocWithFlags((void **)&slot->dev, want, hipDeviceMallocFinegrained));
			slot->host = slot->dev;                                  // one address for the CPU's stores and the kernels' loads
			slot->cap = want;
		}
		used = 0;
		return PCR_OK;
	}
	void seal(uint32_t pass_seq)
	{
		__builtin_ia32_sfence();                                     // the write-combined stores leave the core before the doorbell is rung
		slot->guard_seq = pass_seq + 1;
	}
	int begin(size_t bytes)
	{
		slot = &ctx->stage[ctx->stage_next];
		ctx->stage_next = (ctx->stage_next + 1) % pcr_ctx::STAGE_RING;
		if(slot->busy){ HIP_TRY(hipEventSynchronize(slot->done)); slot->busy = false; }
		if(slot->guard_seq){ const int grc = wait_published(ctx, slot->guard_seq); if(grc != PCR_OK) return grc; slot->guard_seq = 0; }
		if(bytes > slot->cap){
			if(slot->host){ (void)hipHostFree(slot->host); slot->host = nullptr; slot->cap = 0; }
			const size_t want = std::max<size_t>(bytes*2, 1 << 16);
			HIP_TRY(hipHostMalloc((void **)&slot->host, want, hipHostMallocMapped | hipHostMallocCoherent));
			HIP_TRY(hipHostGetDevicePointer((void **)&slot->dev, slot->host, 0));
			slot->cap = want;
		}
		if(!slot->done) HIP_TRY(hipEventCreateWithFlags(&slot->done, hipEventDisableTiming));
		int rc = ctx->arena.ensure(bytes + 256);
		used = 0;
		return rc;
	}
	// returns the DEVICE address the bytes will have
	template<class T> T *put(const T *src, size_t n)
	{
		used = (used + 15) & ~size_t(15);
		memcpy(slot->host + used, src, n*sizeof(T));
		T *dev = (T *)((direct ? slot->dev : ctx->arena.p) + used);
		used += n*sizeof(T);
		return dev;
	}
	// z0/z1: device regions to clear in the same launch (16-byte aligned, sizes rounded UP to 16 bytes: the
	// caller's buffers must be allocated with that slack)
	int ship(void *z0 = nullptr, size_t bytes0 = 0, void *z1 = nullptr, size_t bytes1 = 0, void *z2 = nullptr, size_t bytes2 = 0, uint32_t guard_seq = 0)
	{
		const uint32_t n16 = (uint32_t)((used + 15)/16), n0 = (uint32_t)((bytes0 + 15)/16), n1 = (uint32_t)((bytes1 + 15)/16), n2 = (uint32_t)((bytes2 + 15)/16);
		const uint32_t most = std::max(std::max(n16, n2), std::max(n0, n1));
		if(most){
			const unsigned grid = std::min<unsigned>((most + 255)/256, 512u);
			hipLaunchKernelGGL(k_stage, dim3(grid), dim3(256), 0, ctx->stream, (const uint4 *)slot->dev, (uint4 *)ctx->arena.p, n16,
				(uint4 *)z0, n0, (uint4 *)z1, n1, (uint4 *)z2, n2);
			HIP_TRY(hipGetLastError());
		}
		// The slot may be rewritten once k_stage has run.  A pass publishes its counters after that (mailbox), so
		// its sequence number is the guard; other callers record an event (an event packet between two kernels
		// costs a ~5 us queue bubble, which is why passes avoid it).
		if(guard_seq) slot->guard_seq = guard_seq;
		else{
			HIP_TRY(hipEventRecord(slot->done, ctx->stream));
			slot->busy = true;
		}
		return PCR_OK;
	}
};

// The pass's counters go to the host through mapped memory: no copy-engine packet, no interrupt wake-up.
__global__ void k_publish(const uint32_t *__restrict__ counters, pcr_ctx::Mail *mail, uint32_t seq)
{
	if(threadIdx.x < 4) mail->counters[threadIdx.x] = counters[threadIdx.x];
	__threadfence_system();
	__syncthreads();
	if(threadIdx.x == 0) __hip_atomic_store((uint32_t *)&mail->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

int mail_wait(pcr_ctx *ctx, uint32_t seq, uint32_t out[4])
{
	uint64_t spins = 0;
	pcr_ctx::Mail *const slot = ctx->mail + (seq % pcr_ctx::MAIL_RING);
	while(__atomic_load_n((const uint32_t *)&slot->seq, __ATOMIC_ACQUIRE) != seq){
		__builtin_ia32_pause();
		if((++spins & 0x3FFF) == 0){
			const hipError_t e = hipStreamQuery(ctx->stream);
			if(e == hipSuccess){                          // everything drained: the flag must be there now
				if(__atomic_load_n((const uint32_t *)&slot->seq, __ATOMIC_ACQUIRE) == seq) break;
				g_err = "device pass finished without publishing its counters"; return PCR_ERR_DEVICE;
			}
			if(e != hipErrorNotReady){ g_err = std::string("device pass failed: ") + hipGetErrorString(e); return PCR_ERR_DEVICE; }
		}
	}
	for(int i = 0;i < 4;++i) out[i] = slot->counters[i];
	return PCR_OK;
}

// ---- small results back to the host through mapped memory (see pcr_ctx::ret_host)
__global__ void k_return(const uint4 *__restrict__ s0, uint32_t n0, const uint4 *__restrict__ s1, uint32_t n1,
	const uint4 *__restrict__ s2, uint32_t n2, uint4 *__restrict__ dst)
{
	const uint32_t stride = gridDim.x*blockDim.x, t = blockIdx.x*blockDim.x + threadIdx.x;
	for(uint32_t i = t;i < n0;i += stride) dst[i] = s0[i];
	for(uint32_t i = t;i < n1;i += stride) dst[n0 + i] = s1[i];
	for(uint32_t i = t;i < n2;i += stride) dst[n0 + n1 + i] = s2[i];
}

__global__ void k_return_flag(uint32_t *flag, uint32_t seq)
{
	__threadfence_system();
	__hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Copies up to three device regions (16-byte aligned; sizes are rounded UP to 16 bytes, so the buffers need that
// slack) back to back into the mapped return buffer and waits for them.  out[k] = host address of region k.
int return_to_host(pcr_ctx *ctx, const void *s0, size_t b0, const void *s1, size_t b1, const void *s2, size_t b2, const uint8_t *out[3])
{
	const uint32_t n0 = (uint32_t)((b0 + 15)/16), n1 = (uint32_t)((b1 + 15)/16), n2 = (uint32_t)((b2 + 15)/16);
	const size_t need = ((size_t)n0 + n1 + n2)*16;
	if(!ctx->ret_flag){
		HIP_TRY(hipHostMalloc((void **)&ctx->ret_flag, 64, hipHostMallocMapped | hipHostMallocCoherent));
		HIP_TRY(hipHostGetDevicePointer((void **)&ctx->ret_flag_dev, ctx->ret_flag, 0));
		*ctx->ret_flag = 0;
	}
	if(need > ctx->ret_cap){
		HIP_TRY(hipStreamSynchronize(ctx->stream));
		if(ctx->ret_host){ (void)hipHostFree(ctx->ret_host); ctx->ret_host = nullptr; ctx->ret_cap = 0; }
		const size_t want = std::max<size_t>(need*2, 1 << 20);
		HIP_TRY(hipHostMalloc((void **)&ctx->ret_host, want, hipHostMallocMapped | hipHostMallocCoherent));
		HIP_TRY(hipHostGetDevicePointer((void **)&ctx->ret_dev, ctx->ret_host, 0));
		ctx->ret_cap = want;
	}
	const uint32_t most = std::max(n0, std::max(n1, n2));
	if(most){
		const unsigned grid = std::min<unsigned>((most + 255)/256, 1024u);
		hipLaunchKernelGGL(k_return, dim3(grid), dim3(256), 0, ctx->stream, (const uint4 *)s0, n0, (const uint4 *)s1, n1, (const uint4 *)s2, n2, (uint4 *)ctx->ret_dev);
		HIP_TRY(hipGetLastError());
	}
	const uint32_t seq = ++ctx->ret_seq;
	hipLaunchKernelGGL(k_return_flag, dim3(1), dim3(1), 0, ctx->stream, ctx->ret_flag_dev, seq);
	HIP_TRY(hipGetLastError());
	uint64_t spins = 0;
	while(__atomic_load_n((const uint32_t *)ctx->ret_flag, __ATOMIC_ACQUIRE) != seq){
		__builtin_ia32_pause();
		if((++spins & 0x3FFF) == 0){
			const hipError_t e = hipStreamQuery(ctx->stream);
			if(e == hipSuccess){
				if(__atomic_load_n((const uint32_t *)ctx->ret_flag, __ATOMIC_ACQUIRE) == seq) break;
				g_err = "device work finished without raising the return flag"; return PCR_ERR_DEVICE;
			}
			if(e != hipErrorNotReady){ g_err = std::string("device work failed: ") + hipGetErrorString(e); return PCR_ERR_DEVICE; }
		}
	}
	out[0] = ctx->ret_host; out[1] = ctx->ret_host + (size_t)n0*16; out[2] = ctx->ret_host + ((size_t)n0 + n1)*16;
	return PCR_OK;
}

// Small inputs of a synchronous call: copied into a host-mapped buffer that the kernel reads in place (a pageable
// hipMemcpyAsync costs ~20 us).  Valid until the next call; the caller waits for its results before returning.
int mapped_input(pcr_ctx *ctx, const void *src, size_t bytes, const void **dev_out)
{
	if(bytes > ctx->in_cap){
		HIP_TRY(hipStreamSynchronize(ctx->stream));
		if(ctx->in_host){ (void)hipHostFree(ctx->in_host); ctx->in_host = nullptr; ctx->in_cap = 0; }
		const size_t want = std::max<size_t>(bytes*2, 1 << 18);
		HIP_TRY(hipHostMalloc((void **)&ctx->in_host, want, hipHostMallocMapped | hipHostMallocCoherent));
		HIP_TRY(hipHostGetDevicePointer((void **)&ctx->in_dev, ctx->in_host, 0));
		ctx->in_cap = want;
	}
	memcpy(ctx->in_host, src, bytes);
	*dev_out = ctx->in_dev;
	return PCR_OK;
}

// pcr_screen_device: the amplicon screen's oligo table and the clearing of its result bitsets ride in the
// select pass's staging launch, and the pass's counters are published by k_match instead of k_publish.
struct FusedAmp {
	const pcr_pair *pairs; uint32_t n_pairs; const pcr_amplify_args *a; uint64_t *d_fr, *d_rf;
	bool staged = false; const OligoDev *d_oligos = nullptr; uint32_t pub_seq = 0; const uint32_t *pub_counters = nullptr;
	bool posted = false;       // k_post ran the whole tail (amplicon screen included)
};

void build_oligos(const pcr_pair *pairs, uint32_t n_pairs, const pcr_amplify_args *a, std::vector<OligoDev> &ol)
{
	const float thr2 = a->collect_threshold*a->collect_threshold;                // pcr_assay.cpp:31-32
	ol.resize(2*(size_t)n_pairs);
	for(uint32_t i = 0;i < n_pairs;++i){
		fill_oligo(ol[2*i], pairs[i].f.w, thr2);
		fill_oligo(ol[2*i + 1], pairs[i].r.w, thr2);
	}
}

// The fused tail (k_post) leaves the DB in place but builds no touched list: do it for the callers that walk it.
int ensure_touched(pcr_ctx *ctx, SeqSet &S)
{
	if(S.touched_built || !S.have_db || !S.ctrl.p) return PCR_OK;
	if(S.touched_from_seg) hipLaunchKernelGGL(k_touched_seg, dim3((S.n + 255)/256), dim3(256), 0, ctx->stream, S.d_seg_hi, S.n, S.ctrl.p, S.touched.p);
	else hipLaunchKernelGGL(k_touched, dim3((S.n + 255)/256), dim3(256), 0, ctx->stream, S.ctrl.p + 8, S.n, S.ctrl.p, S.touched.p);
	HIP_TRY(hipGetLastError());
	uint32_t n = 0;
	HIP_TRY(hipMemcpyAsync(&n, S.ctrl.p + 3, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(hipStreamSynchronize(ctx->stream));
	S.n_touched = n; S.n_entries = n ? 1 : 0; S.touched_built = true;
	return PCR_OK;
}

int amplify_launch(pcr_ctx *ctx, SeqSet &S, const pcr_pair *pairs, uint32_t n_pairs, const pcr_amplify_args *a,
	uint64_t *d_fr, uint64_t *d_rf, const FusedAmp *fa = nullptr)
{
	{ const int erc = ensure_touched(ctx, S); if(erc != PCR_OK) return erc; }
	if(!S.have_db){ g_err = "pcr_amplify: no word DB (call pcr_select_words first)"; return PCR_ERR_STATE; }
	const uint64_t words = (S.n + 63)/64;
	const size_t bits_bytes = (size_t)n_pairs*words*sizeof(uint64_t);
	if(S.n_entries == 0 || n_pairs == 0){
		if(bits_bytes){
			HIP_TRY(hipMemsetAsync(d_fr, 0, bits_bytes, ctx->stream));
			HIP_TRY(hipMemsetAsync(d_rf, 0, bits_bytes, ctx->stream));
		}
		return PCR_OK;
	}
	HostTimer timer(ctx, 4);
	int rc;
	const bool prestaged = fa && fa->staged;
	if(prestaged) ctx->d_oligos = fa->d_oligos;
	else{
		std::vector<OligoDev> ol;
		build_oligos(pairs, n_pairs, a, ol);
		Stager st(ctx);
		if((rc = st.begin(ol.size()*sizeof(OligoDev) + 64)) != PCR_OK) return rc;
		ctx->d_oligos = st.put(ol.data(), ol.size());
		// the result bitsets are cleared by the staging kernel when they allow 16-byte stores
		const bool vec_ok = ((uintptr_t)d_fr % 16 == 0) && ((uintptr_t)d_rf % 16 == 0) && (bits_bytes % 16 == 0);
		if(!vec_ok){
			HIP_TRY(hipMemsetAsync(d_fr, 0, bits_bytes, ctx->stream));
			HIP_TRY(hipMemsetAsync(d_rf, 0, bits_bytes, ctx->stream));
		}
		if((rc = vec_ok ? st.ship(d_fr, bits_bytes, d_rf, bits_bytes) : st.ship()) != PCR_OK) return rc;
	}
	timer.next(5);
	const uint32_t mask_words = (2*n_pairs + 31)/32;
	if((rc = ctx->mask.ensure((size_t)S.n_slots*mask_words)) != PCR_OK) return rc;
	if((rc = ctx->status.ensure(1)) != PCR_OK) return rc;
	// the number of touched sequences: known to the host after a synchronous select, read on the device
	// (conservative grids, grid-stride loops) after an asynchronous one
	const bool dev_n = S.n_touched == N_TOUCHED_UNKNOWN;
	const uint32_t *n_dev = dev_n ? S.ctrl.p + 3 : nullptr;
	const uint32_t n_t = dev_n ? 0u : S.n_touched, n_db = n_t*S.db_cap;
	const unsigned threads = dev_n ? 256 : 128;
	const unsigned grid = dev_n ? (unsigned)std::min<uint64_t>(((uint64_t)S.n*S.db_cap + threads - 1)/threads, 1024) : (n_db + threads - 1)/threads;
	const unsigned mgrid = dev_n ? std::min<unsigned>((S.n + MATCH_WAVES - 1)/MATCH_WAVES, 1024u) : (n_t + MATCH_WAVES - 1)/MATCH_WAVES;
	// k_match also clears the status word (it runs before k_pair in stream order)
	hipLaunchKernelGGL(k_match, dim3(std::max(mgrid, 1u)), dim3(64*MATCH_WAVES), 0, ctx->stream, S.db.p, n_t, n_dev, S.db_cap,
		S.touched.p, S.d_seg_hi, ctx->d_oligos, 2*n_pairs, mask_words, ctx->mask.p, ctx->status.p,
		prestaged ? fa->pub_counters : (const uint32_t *)nullptr, prestaged ? ctx->mail_dev + (fa->pub_seq % pcr_ctx::MAIL_RING) : (pcr_ctx::Mail *)nullptr,
		prestaged ? fa->pub_seq : 0u);
	HIP_TRY(hipGetLastError());
	hipLaunchKernelGGL(k_pair, dim3(grid), dim3(threads), 0, ctx->stream, S.db.p, n_db, n_dev, S.db_cap, S.touched.p, S.d_seg_hi, ctx->mask.p,
		mask_words, ctx->d_oligos, n_pairs, S.planes.p, S.d_blk_off.p, S.d_len.p, S.d_active.p, a->amp_min, a->amp_max,
		a->ident_threshold, a->use_taq_mama, d_fr, d_rf, words, ctx->status.p);
	HIP_TRY(hipGetLastError());
	return PCR_OK;
}

template<int NSLOT, int KLO>
int launch_scan2_nw(pcr_ctx *ctx, SeqSet &S, uint32_t nw, uint32_t group0, uint32_t n_groups, uint32_t gw, uint32_t ncand, const HitSink &sink,
	const uint32_t *d_tab, const uint32_t *d_bias, const uint32_t *tile_ids, uint32_t n_tiles, const uint32_t *orient_ids)
{
#define SCAN2_ARGS S.nib.p, S.planes.p, S.valid_d(), S.d_blk_off.p, S.d_len.p, S.d_active.p, S.tile_seq.p, S.tile_pos0.p, \
	d_tab, d_bias, group0, gw, tile_ids, orient_ids, ctx->d_cand_fwd, ctx->d_cand_rc, ctx->d_cand_floor, ncand, sink
	const dim3 grid(n_tiles, n_groups), block(SCAN2_THREADS);
	switch(nw){
		case 1: hipLaunchKernelGGL((k_scan2<1, NSLOT, KLO>), grid, block, 0, ctx->stream, SCAN2_ARGS); break;
		case 2: hipLaunchKernelGGL((k_scan2<2, NSLOT, KLO>), grid, block, 0, ctx->stream, SCAN2_ARGS); break;
		case 3: hipLaunchKernelGGL((k_scan2<3, NSLOT, KLO>), grid, block, 0, ctx->stream, SCAN2_ARGS); break;
		case 4: hipLaunchKernelGGL((k_scan2<4, NSLOT, KLO>), grid, block, 0, ctx->stream, SCAN2_ARGS); break;
		case 5: hipLaunchKernelGGL((k_scan2<5, NSLOT, KLO>), grid, block, 0, ctx->stream, SCAN2_ARGS); break;
		case 6: hipLaunchKernelGGL((k_scan2<6, NSLOT, KLO>), grid, block, 0, ctx->stream, SCAN2_ARGS); break;
		case 7: hipLaunchKernelGGL((k_scan2<7, NSLOT, KLO>), grid, block, 0, ctx->stream, SCAN2_ARGS); break;
		default: hipLaunchKernelGGL((k_scan2<8, NSLOT, KLO>), grid, block, 0, ctx->stream, SCAN2_ARGS); break;
	}
#undef SCAN2_ARGS
	HIP_TRY(hipGetLastError());
	return PCR_OK;
}

// The bit-sliced scan over all orientation groups: full groups (8 words) in one launch, the
// partial last group in a second one.
int launch_scan2(pcr_ctx *ctx, SeqSet &S, const Scan2Tables &T, uint32_t ncand, const HitSink &sink,
	const uint32_t *d_tab, const uint32_t *d_bias, const uint32_t *tile_ids, uint32_t n_tiles, const uint32_t *orient_ids)
{
	if(n_tiles == 0 || T.n_groups == 0) return PCR_OK;
	const uint32_t full = (T.last_words == T.gw) ? T.n_groups : T.n_groups - 1;
	int rc = PCR_OK;
	if(full){
		rc = (T.nslot == 26) ? launch_scan2_nw<26, 3>(ctx, S, T.gw, 0, full, T.gw, ncand, sink, d_tab, d_bias, tile_ids, n_tiles, orient_ids)
			: launch_scan2_nw<32, 0>(ctx, S, T.gw, 0, full, T.gw, ncand, sink, d_tab, d_bias, tile_ids, n_tiles, orient_ids);
		if(rc != PCR_OK) return rc;
	}
	if(full < T.n_groups){
		rc = (T.nslot == 26) ? launch_scan2_nw<26, 3>(ctx, S, T.last_words, full, 1, T.gw, ncand, sink, d_tab, d_bias, tile_ids, n_tiles, orient_ids)
			: launch_scan2_nw<32, 0>(ctx, S, T.last_words, full, 1, T.gw, ncand, sink, d_tab, d_bias, tile_ids, n_tiles, orient_ids);
	}
	return rc;
}

// Exact DB size: sum of the per-sequence fills (done on the host on request; a device-side
// same-address atomic per sequence would serialise, ~35 ns each).
int count_entries(pcr_ctx *ctx, SeqSet &S)
{
	uint64_t total = 0;
	if(S.n_touched){
		std::vector<uint32_t> hi(S.n);
		HIP_TRY(hipMemcpyAsync(hi.data(), S.d_seg_hi, S.n*sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
		HIP_TRY(hipStreamSynchronize(ctx->stream));
		for(uint32_t q = 0;q < S.n;++q){ if(hi[q]) total += hi[q] - (uint64_t)q*S.db_cap; }
	}
	S.n_entries = (uint32_t)total;
	return PCR_OK;
}

} // namespace

// ============================================================================== C-ABI
extern "C" {

const char *pcr_last_error(void) { return g_err.c_str(); }

pcr_ctx *pcr_create(int device, void *hip_stream, const pcr_params *params)
{
	int count = 0;
	if(hipGetDeviceCount(&count) != hipSuccess || count <= 0){
		g_err = "pcr_create: no HIP device available (this library has no CPU fallback)";
		return nullptr;
	}
	if(device < 0 || device >= count){ g_err = "pcr_create: bad device index"; return nullptr; }
	if(hipSetDevice(device) != hipSuccess){ g_err = "pcr_create: hipSetDevice failed"; return nullptr; }
	hipDeviceProp_t prop;
	if(hipGetDeviceProperties(&prop, device) != hipSuccess){ g_err = "pcr_create: hipGetDeviceProperties failed"; return nullptr; }
	if(std::string(prop.gcnArchName).find("gfx950") == std::string::npos){
		g_err = std::string("pcr_create: device is ") + prop.gcnArchName + ", this build targets gfx950 only";
		return nullptr;
	}
	pcr_ctx *ctx = new pcr_ctx();
	ctx->device = device;
	if(hip_stream){ ctx->stream = (hipStream_t)hip_stream; ctx->own_stream = false; }
	else{
		if(hipStreamCreate(&ctx->stream) != hipSuccess){ g_err = "pcr_create: hipStreamCreate failed"; delete ctx; return nullptr; }
		ctx->own_stream = true;
	}
	if(params){ ctx->params = *params; }
	else{ ctx->params.pack_max_degen = 256; ctx->params.pack_min_gc = 0.0f; ctx->params.pack_max_gc = 1.0f; }
	{ hipDeviceProp_t prop; if(hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) ctx->n_cu = (uint32_t)prop.multiProcessorCount; }
	if(const char *v = getenv("PCRAMP_TIMING")) ctx->timing = v[0] == '1';
	ctx->debug_log = getenv("PCRAMP_DEBUG") != nullptr;
	if(const char *v = getenv("PCRAMP_DEBUG_EPOCH")) ctx->debug_epoch = (uint32_t)strtoul(v, nullptr, 0);   // test hook: start the pass counter near its wrap
	if(const char *v = getenv("PCRAMP_OPT_PM")) ctx->opt_pm_global = v[0] == 'g';
	if(const char *v = getenv("PCRAMP_DEBUG_OPT_TASKS")){ unsigned a = 0, b = 0; if(sscanf(v, "%u,%u", &a, &b) >= 1){ ctx->opt_dbg_wg_tasks = a; ctx->opt_dbg_task_cap = b; } }
	if(const char *v = getenv("PCRAMP_SEED")) ctx->force_seed1 = v[0] == '1';
	if(const char *v = getenv("PCRAMP_SEED_TABLES")) ctx->host_seed_tables = v[0] == 'h';
	if(const char *v = getenv("PCRAMP_S2DBG")) ctx->s2_dbg = (uint32_t)atoi(v);
	if(const char *v = getenv("PCRAMP_IRR_INDEX")) ctx->no_irr_index = v[0] == '0';
	if(const char *v = getenv("PCRAMP_SEED3")) ctx->no_seed3 = v[0] == '0';
	if(const char *v = getenv("PCRAMP_SCAN")){ if(v[0] == '1') ctx->scan_version = 1; else if(v[0] == '2') ctx->scan_version = 2; }   // A/B: 1 = popcount scan, 2 = bit-sliced only
	{
		// direct staging: fine-grained device memory the CPU can store into (large BAR) and whose stores a later launch sees.
		// Probed, not assumed: the device must report a large BAR (otherwise a CPU store into the allocation would fault), and a
		// pattern stored by the CPU, fenced, must come back through a kernel that copies it into mapped host memory.
		const char *v = getenv("PCRAMP_STAGE");
		if(!(v && v[0] == 'k') && prop.isLargeBar){
			void *probe = nullptr; uint32_t *back = nullptr, *back_dev = nullptr;
			constexpr uint32_t PROBE_WORDS = 1024;
			if(hipExtMallocWithFlags(&probe, PROBE_WORDS*sizeof(uint32_t), hipDeviceMallocFinegrained) == hipSuccess && probe &&
			   hipHostMalloc((void **)&back, PROBE_WORDS*sizeof(uint32_t), hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess && back &&
			   hipHostGetDevicePointer((void **)&back_dev, back, 0) == hipSuccess){
				hipPointerAttribute_t at;
				bool ok = hipPointerGetAttributes(&at, probe) == hipSuccess;
				for(int round = 0;ok && round < 2;++round){
					volatile uint32_t *w = (volatile uint32_t *)probe;
					for(uint32_t i = 0;i < PROBE_WORDS;++i){ w[i] = 0x9E3779B9u*(i + 1) + (uint32_t)round; back[i] = 0; }
					__builtin_ia32_sfence();
					hipLaunchKernelGGL(k_stage, dim3(1), dim3(256), 0, ctx->stream, (const uint4 *)probe, (uint4 *)back_dev, PROBE_WORDS/4,
						(uint4 *)nullptr, 0u, (uint4 *)nullptr, 0u, (uint4 *)nullptr, 0u);
					ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(ctx->stream) == hipSuccess;
					for(uint32_t i = 0;ok && i < PROBE_WORDS;++i) ok = back[i] == 0x9E3779B9u*(i + 1) + (uint32_t)round;
				}
				ctx->direct_ok = ok;
			}
			else (void)hipGetLastError();
			if(probe) (void)hipFree(probe);
			if(back) (void)hipHostFree(back);
		}
	}
	ctx->filt.max_degen = ctx->params.pack_max_degen;
	ctx->filt.set_gc(ctx->params.pack_min_gc, ctx->params.pack_max_gc);
	if(hipMemcpyToSymbol(HIP_SYMBOL(c_taq_mama), h_taq_mama, sizeof(h_taq_mama)) != hipSuccess ||
	   hipMemcpyToSymbol(HIP_SYMBOL(thermo::c_H), PCR_NN_H, sizeof(PCR_NN_H)) != hipSuccess ||
	   hipMemcpyToSymbol(HIP_SYMBOL(thermo::c_S), PCR_NN_S, sizeof(PCR_NN_S)) != hipSuccess ||
	   hipMemcpyToSymbol(HIP_SYMBOL(thermo::c_loop_S), PCR_LOOP_S, sizeof(PCR_LOOP_S)) != hipSuccess ||
	   hipMemcpyToSymbol(HIP_SYMBOL(thermo::c_bulge_S), PCR_BULGE_S, sizeof(PCR_BULGE_S)) != hipSuccess ||
	   hipMemcpyToSymbol(HIP_SYMBOL(thermo::c_hairpin_S), PCR_HAIRPIN_S, sizeof(PCR_HAIRPIN_S)) != hipSuccess ||
	   hipMemcpyToSymbol(HIP_SYMBOL(thermo::c_special_H), PCR_SPECIAL_H, sizeof(PCR_SPECIAL_H)) != hipSuccess ||
	   hipMemcpyToSymbol(HIP_SYMBOL(thermo::c_special_S), PCR_SPECIAL_S, sizeof(PCR_SPECIAL_S)) != hipSuccess ||
	   hipMemcpyToSymbol(HIP_SYMBOL(thermo::c_special_key), PCR_SPECIAL_KEY, sizeof(PCR_SPECIAL_KEY)) != hipSuccess ||
	   hipMemcpyToSymbol(HIP_SYMBOL(thermo::c_wc), PCR_WC, sizeof(PCR_WC)) != hipSuccess){
		g_err = "pcr_create: hipMemcpyToSymbol failed"; delete ctx; return nullptr;
	}
	if(hipHostMalloc((void **)&ctx->mail, sizeof(pcr_ctx::Mail)*pcr_ctx::MAIL_RING, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
	   hipHostGetDevicePointer((void **)&ctx->mail_dev, ctx->mail, 0) != hipSuccess){
		g_err = "pcr_create: mapped host allocation failed"; if(ctx->mail) (void)hipHostFree(ctx->mail); delete ctx; return nullptr;
	}
	memset((void *)ctx->mail, 0, sizeof(pcr_ctx::Mail)*pcr_ctx::MAIL_RING);
	return ctx;
}

void pcr_destroy(pcr_ctx *ctx)
{
	if(!ctx) return;
	(void)hipSetDevice(ctx->device);
	(void)hipStreamSynchronize(ctx->stream);
	ctx->pending.clear();
	if(ctx->timing && ctx->n_timed){
		const double n = (double)ctx->n_timed;
		fprintf(stderr, "[pcramp] host us/pass: plan %.1f  stage %.1f  launch %.1f  wait %.1f | amplify prep %.1f  launch %.1f  (%llu passes)\n",
			ctx->t_host[0]/n, ctx->t_host[1]/n, ctx->t_host[2]/n, ctx->t_host[3]/n, ctx->t_host[4]/n, ctx->t_host[5]/n, (unsigned long long)ctx->n_timed);
	}
	for(auto &pr : ctx->prof_events){ (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
	for(int k = 0;k < PCR_PROF_KERNELS;++k){ for(auto &pr : ctx->prof_events_k[k]){ (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); } }
	for(int s = 0;s < PCR_N_SETS;++s) ctx->sets[s].release();
	ctx->best.release();
	ctx->counters.release(); ctx->mask.release(); ctx->status.release(); ctx->hits.release();
	ctx->s1_image.release(); ctx->s1_heads.release(); ctx->s1_multi.release(); ctx->s1_part.release();
	ctx->bits_fr.release(); ctx->bits_rf.release(); ctx->arena.release(); ctx->fin_scratch.release();
	for(auto &sl : ctx->stage){ if(sl.host) (void)hipHostFree(sl.host); if(sl.done) (void)hipEventDestroy(sl.done); }
	for(auto &sl : ctx->dstage){ if(sl.dev) (void)hipFree(sl.dev); }
	if(ctx->mail) (void)hipHostFree(ctx->mail);
	if(ctx->ret_host) (void)hipHostFree(ctx->ret_host);
	if(ctx->in_host) (void)hipHostFree(ctx->in_host);
	if(ctx->ret_flag) (void)hipHostFree(ctx->ret_flag);
	if(ctx->sw_pin) (void)hipHostFree(ctx->sw_pin);
	for(int k = 0;k < 2;++k){ if(ctx->sw_done[k]) (void)hipEventDestroy(ctx->sw_done[k]); }
	ctx->oligos.release(); ctx->sw_jobs.release(); ctx->sw_out.release(); ctx->sw_q.release(); ctx->sw_qlen.release(); ctx->sw_t.release(); ctx->entry_codes.release(); ctx->entry_lens.release(); ctx->amp_recs.release(); ctx->amp_recs2.release(); ctx->amp_keys.release(); ctx->amp_pkeys.release(); ctx->amp_pair_start.release(); ctx->sort_tmp.release(); ctx->bg_pairs.release(); ctx->th_jobs.release(); ctx->th_out.release(); ctx->th_dg.release(); ctx->th_dbg.release(); ctx->split_where.release(); ctx->th_map.release(); ctx->th_bad.release(); ctx->mx_keys.release(); ctx->mx_count.release(); ctx->mx_amp.release(); ctx->opt_oligos.release(); ctx->opt_jobs.release(); ctx->opt_cov.release(); ctx->opt_loc.release(); ctx->opt_tasks.release();
	if(ctx->aux_stream){ (void)hipStreamSynchronize(ctx->aux_stream); (void)hipStreamDestroy(ctx->aux_stream); }
	if(ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
	delete ctx;
}

uint32_t pcr_num_sequences(pcr_ctx *ctx, pcr_set which) { return (ctx && set_ok(which)) ? ctx->sets[which].n : 0; }
uint64_t pcr_bitset_words(pcr_ctx *ctx, pcr_set which) { return (ctx && set_ok(which)) ? (ctx->sets[which].n + 63)/64 : 0; }

int pcr_staging_mode(pcr_ctx *ctx) { return (ctx && ctx->direct_ok) ? 1 : 0; }

int pcr_synchronize(pcr_ctx *ctx)
{
	if(!ctx){ g_err = "null ctx"; return PCR_ERR_ARG; }
	HIP_TRY(hipSetDevice(ctx->device));
	DRAIN(ctx);
	HIP_TRY(hipStreamSynchronize(ctx->stream));
	return PCR_OK;
}

static int load_sequences_impl(pcr_ctx *ctx, int which, const uint8_t *packed4, const uint64_t *byte_offsets,
	const uint64_t *lengths, const float *weights, uint32_t n);

int pcr_load_sequences(pcr_ctx *ctx, pcr_set which, const uint8_t *packed4, const uint64_t *byte_offsets,
	const uint64_t *lengths, const float *weights, uint32_t n)
{
	if(!set_ok(which)){ g_err = "pcr_load_sequences: unknown sequence set"; return PCR_ERR_ARG; }
	return load_sequences_impl(ctx, (int)which, packed4, byte_offsets, lengths, weights, n);
}

// (which may be the internal scratch set)
static int load_sequences_impl(pcr_ctx *ctx, int which, const uint8_t *packed4, const uint64_t *byte_offsets,
	const uint64_t *lengths, const float *weights, uint32_t n)
{
	if(!ctx || which < 0 || which >= PCR_N_SETS || (n && (!packed4 || !byte_offsets || !lengths))){
		g_err = "pcr_load_sequences: bad argument"; return PCR_ERR_ARG;
	}
	DRAIN(ctx);
	if(n >= (1u << 24)){ g_err = "pcr_load_sequences: at most 2^24-1 sequences per GPU shard"; return PCR_ERR_CAPACITY; }
	HIP_TRY(hipSetDevice(ctx->device));
	// a pass publishes its mailbox before its last kernel has finished (k_post publishes at its start): the blocking
	// copies below go through the null stream, which a caller-supplied non-blocking stream does not order against
	HIP_TRY(hipStreamSynchronize(ctx->stream));
	SeqSet &S = ctx->sets[which];
	S.have_db = false; S.n_entries = 0; S.have_codes = false;
	S.n = n;
	S.packed.assign(n, std::vector<uint8_t>());
	S.len.assign(lengths, lengths + n);
	S.weight.assign(n, 1.0f);
	if(weights) S.weight.assign(weights, weights + n);
	S.weight_dirty = true;
	S.ctrl_clean = false; S.touched_from_seg = false;
	S.active.assign(n, 1);
	S.has_eos.assign(n, 0);
	S.blk_off.assign(n + 1, 0);
	S.nblk_real.assign(n, 0);
	S.irr_host.assign(n, std::vector<pcrhost::IrrEntry>());
	uint64_t total_bytes = 0, total_blocks = 0, n_tiles = 0;
	std::vector<uint64_t> dev_byte_off(n);
	for(uint32_t s = 0;s < n;++s){
		if(lengths[s] >= (uint64_t(1) << 31)){ g_err = "pcr_load_sequences: sequence longer than 2^31-1 bases"; return PCR_ERR_CAPACITY; }
		const uint64_t nb = (lengths[s] + 1)/2;
		S.packed[s].assign(packed4 + byte_offsets[s], packed4 + byte_offsets[s] + nb);
		if((lengths[s] & 1) && nb) S.packed[s][nb - 1] &= 0xF0;                  // pad nibble = EOS (sequence.cpp:21)
		{
			const std::vector<uint8_t> &b = S.packed[s];
			bool eos = false;
			const uint64_t full = lengths[s]/2;
			for(uint64_t k = 0;k < full && !eos;++k) eos = ((b[k] & 0xF0) == 0) || ((b[k] & 0x0F) == 0);
			if(!eos && (lengths[s] & 1)) eos = ((b[full] & 0xF0) == 0);
			S.has_eos[s] = eos ? 1 : 0;
		}
		dev_byte_off[s] = total_bytes;
		total_bytes += nb;
		S.blk_off[s] = total_blocks;
		S.nblk_real[s] = (lengths[s] + 31)/32;
		total_blocks += S.nblk_real[s] + 2;
		if(lengths[s] >= 32) n_tiles += (lengths[s] - 7 + TILE_POS - 1)/TILE_POS;   // tiles cover window starts 0..L-32 AND seed positions 0..L-8 (k_seed)
	}
	S.blk_off[n] = total_blocks;
	S.total_blocks = total_blocks;
	if(n_tiles >= (uint64_t(1) << 31)){ g_err = "pcr_load_sequences: too many tiles"; return PCR_ERR_CAPACITY; }
	S.n_tiles = (uint32_t)n_tiles;

	// host-side index pieces: block -> sequence map, tile list, irregular words
	std::vector<uint32_t> blk_seq(total_blocks), tile_seq(n_tiles), tile_pos0(n_tiles);
	uint64_t t = 0;
	S.pix_valid = false; S.pix_usable = false;
	for(uint32_t s = 0;s < n;++s){
		for(uint64_t b = S.blk_off[s];b < S.blk_off[s + 1];++b) blk_seq[b] = s;
		if(lengths[s] >= 32){
			const uint64_t nt = (lengths[s] - 7 + TILE_POS - 1)/TILE_POS;
			for(uint64_t k = 0;k < nt;++k, ++t){ tile_seq[t] = s; tile_pos0[t] = (uint32_t)(k*TILE_POS); }
		}
		pcrhost::PackedSeq q; q.buf = S.packed[s].data(); q.len = lengths[s];
		if(!pcrhost::irregular_words(q, ctx->filt, S.irr_host[s])){
			g_err = "pcr_load_sequences: more than 64 irregular words share one (loc, strand)"; return PCR_ERR_CAPACITY;
		}
	}

	int rc;
	DevBuf<uint8_t> d_packed; DevBuf<uint64_t> d_byte_off;
	if((rc = d_packed.ensure(total_bytes)) != PCR_OK) return rc;
	if((rc = d_byte_off.ensure(n)) != PCR_OK){ d_packed.release(); return rc; }
	auto fail = [&](int code){ d_packed.release(); d_byte_off.release(); return code; };
	if((rc = S.planes.ensure(total_blocks)) != PCR_OK) return fail(rc);
	if((rc = S.valid.ensure(S2_VALID_FRONT + total_blocks + S2_TAIL_BLOCKS)) != PCR_OK) return fail(rc);
	if((rc = S.nib.ensure(total_blocks*4 + 8)) != PCR_OK) return fail(rc);
	if((rc = S.tb.ensure(S2_TB_FRONT + (total_blocks + S2_TAIL_BLOCKS)*2 + 8)) != PCR_OK) return fail(rc);
	if((rc = S.tile_degen.ensure(n_tiles + 1)) != PCR_OK) return fail(rc);
	if((rc = S.degen_tiles.ensure(n_tiles + 1)) != PCR_OK) return fail(rc);
	if((rc = S.blk_seq.ensure(total_blocks)) != PCR_OK) return fail(rc);
	if((rc = S.tile_seq.ensure(n_tiles)) != PCR_OK) return fail(rc);
	if((rc = S.tile_pos0.ensure(n_tiles)) != PCR_OK) return fail(rc);
	if((rc = S.d_len.ensure(n)) != PCR_OK) return fail(rc);
	if((rc = S.d_blk_off.ensure(n + 1)) != PCR_OK) return fail(rc);
	if((rc = S.d_nblk_real.ensure(n)) != PCR_OK) return fail(rc);
	if((rc = S.d_active.ensure(n)) != PCR_OK) return fail(rc);
	if((rc = S.d_has_eos.ensure(n)) != PCR_OK) return fail(rc);
#define H2D(dst, src, bytes) do{ if((bytes) > 0){ hipError_t e_ = hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice); \
	if(e_ != hipSuccess){ g_err = std::string("hipMemcpy: ") + hipGetErrorString(e_); return fail(PCR_ERR_DEVICE); } } }while(0)
	{   // one transfer for all sequences (a copy per sequence was 10 000 copy-engine packets at C2)
		std::vector<uint8_t> flat(total_bytes);
		for(uint32_t s = 0;s < n;++s){ if(!S.packed[s].empty()) memcpy(flat.data() + dev_byte_off[s], S.packed[s].data(), S.packed[s].size()); }
		H2D(d_packed.p, flat.data(), flat.size());
	}
	H2D(d_byte_off.p, dev_byte_off.data(), n*sizeof(uint64_t));
	H2D(S.blk_seq.p, blk_seq.data(), total_blocks*sizeof(uint32_t));
	H2D(S.tile_seq.p, tile_seq.data(), n_tiles*sizeof(uint32_t));
	H2D(S.tile_pos0.p, tile_pos0.data(), n_tiles*sizeof(uint32_t));
	H2D(S.d_len.p, S.len.data(), n*sizeof(uint64_t));
	H2D(S.d_blk_off.p, S.blk_off.data(), (n + 1)*sizeof(uint64_t));
	H2D(S.d_nblk_real.p, S.nblk_real.data(), n*sizeof(uint64_t));
	H2D(S.d_active.p, S.active.data(), n);
	H2D(S.d_has_eos.p, S.has_eos.data(), n);
#undef H2D
	{   // the padding in front of and behind valid[] / tb[] (what k_seed2's unchecked tile fetch may touch beyond the sequences)
		hipError_t e_ = hipMemsetAsync(S.valid.p, 0, S2_VALID_FRONT*sizeof(uint32_t), ctx->stream);
		if(e_ == hipSuccess) e_ = hipMemsetAsync(S.valid_d() + total_blocks, 0, S2_TAIL_BLOCKS*sizeof(uint32_t), ctx->stream);
		if(e_ == hipSuccess) e_ = hipMemsetAsync(S.tb.p, 0, S2_TB_FRONT*sizeof(uint32_t), ctx->stream);
		if(e_ == hipSuccess) e_ = hipMemsetAsync(S.tb_d() + total_blocks*2, 0, (S2_TAIL_BLOCKS*2 + 8)*sizeof(uint32_t), ctx->stream);
		if(e_ != hipSuccess){ g_err = std::string("load: padding memset: ") + hipGetErrorString(e_); return fail(PCR_ERR_DEVICE); }
	}
	if(total_blocks){
		const unsigned threads = 256;
		const unsigned grid = (unsigned)((total_blocks + threads - 1)/threads);
		hipLaunchKernelGGL(k_transpose, dim3(grid), dim3(threads), 0, ctx->stream, d_packed.p, d_byte_off.p, S.d_len.p,
			S.d_blk_off.p, S.blk_seq.p, S.planes.p, S.nib.p, S.tb_d(), total_blocks);
		if(hipGetLastError() != hipSuccess){ g_err = "k_transpose launch failed"; return fail(PCR_ERR_DEVICE); }
		if((rc = run_valid(ctx, S, 0, total_blocks)) != PCR_OK) return fail(rc);
	}
	S.bucket_cap = 64;               // grown by earlier passes over other data: start small again
	S.n_degen_tiles = 0;
	if(n_tiles){
		hipLaunchKernelGGL(k_tile_degen, dim3((unsigned)((n_tiles + 255)/256)), dim3(256), 0, ctx->stream, S.planes.p, S.d_blk_off.p,
			S.d_nblk_real.p, S.tile_seq.p, S.tile_pos0.p, (uint32_t)n_tiles, S.tile_degen.p);
		if(hipGetLastError() != hipSuccess){ g_err = "k_tile_degen launch failed"; return fail(PCR_ERR_DEVICE); }
	}
	if(total_blocks >= (uint64_t(1) << 40)){ g_err = "pcr_load_sequences: more than 2^40 blocks"; return fail(PCR_ERR_CAPACITY); }
	if((rc = build_tile_desc(ctx, S)) != PCR_OK) return fail(rc);
	if(hipStreamSynchronize(ctx->stream) != hipSuccess){ g_err = "load: stream sync failed"; return fail(PCR_ERR_DEVICE); }
	if(n_tiles){
		std::vector<uint8_t> flags(n_tiles);
		std::vector<uint32_t> list;
		if(hipMemcpy(flags.data(), S.tile_degen.p, n_tiles, hipMemcpyDeviceToHost) != hipSuccess){ g_err = "load: flag download failed"; return fail(PCR_ERR_DEVICE); }
		for(uint64_t t2 = 0;t2 < n_tiles;++t2){ if(flags[t2]) list.push_back((uint32_t)t2); }
		S.n_degen_tiles = (uint32_t)list.size();
		if(!list.empty() && hipMemcpy(S.degen_tiles.p, list.data(), list.size()*sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess){
			g_err = "load: tile list upload failed"; return fail(PCR_ERR_DEVICE);
		}
	}
	d_packed.release(); d_byte_off.release();
	return upload_irregular(ctx, S);
}

int pcr_set_active(pcr_ctx *ctx, pcr_set which, const uint8_t *active)
{
	if(!set_ok(which)){ g_err = "pcr_set_active: unknown sequence set"; return PCR_ERR_ARG; }
	if(!ctx || !active){ g_err = "pcr_set_active: bad argument"; return PCR_ERR_ARG; }
	DRAIN(ctx);
	HIP_TRY(hipSetDevice(ctx->device));
	SeqSet &S = ctx->sets[which];
	for(uint32_t i = 0;i < S.n;++i) S.active[i] = active[i] ? 1 : 0;
	if(S.n){
		HIP_TRY(hipMemcpyAsync(S.d_active.p, S.active.data(), S.n, hipMemcpyHostToDevice, ctx->stream));
		{ const int rc = build_tile_desc(ctx, S); if(rc != PCR_OK) return rc; }
		if(S.pix_valid && S.pix_usable && S.total_blocks){                       // the third form reads the flag out of its per-block words
			hipLaunchKernelGGL(k_blk_active, dim3((unsigned)((S.total_blocks + 255)/256)), dim3(256), 0, ctx->stream, S.blk_info.p, S.blk_seq.p, S.d_active.p, S.total_blocks);
			HIP_TRY(hipGetLastError());
		}
		HIP_TRY(hipStreamSynchronize(ctx->stream));
	}
	return PCR_OK;
}

int pcr_split_many(pcr_ctx *ctx, pcr_set which, const uint32_t *seq, const uint64_t *pos, uint32_t n)
{
	if(!set_ok(which)){ g_err = "pcr_split: unknown sequence set"; return PCR_ERR_ARG; }
	if(!ctx || (n && (!seq || !pos))){ g_err = "pcr_split: bad argument"; return PCR_ERR_ARG; }
	if(n == 0) return PCR_OK;
	DRAIN(ctx);
	HIP_TRY(hipSetDevice(ctx->device));
	SeqSet &S = ctx->sets[which];
	for(uint32_t i = 0;i < n;++i){ if(seq[i] >= S.n || pos[i] >= S.len[seq[i]]){ g_err = "pcr_split: out of range"; return PCR_ERR_ARG; } }
	// host copy, and what the device has to change: per split the block of the plane store and the base's bit in it
	std::vector<uint64_t> where(2*(size_t)n);
	std::vector<uint32_t> touched_seq;
	bool eos_changed = false;
	for(uint32_t i = 0;i < n;++i){
		uint8_t &v = S.packed[seq[i]][pos[i] >> 1];
		v = (pos[i] & 1) ? (v & 0xF0) : (v & 0x0F);                                // sequence.h:232-241
		if(!S.has_eos[seq[i]]){ S.has_eos[seq[i]] = 1; eos_changed = true; }
		where[2*(size_t)i] = S.blk_off[seq[i]] + (pos[i] >> 5); where[2*(size_t)i + 1] = pos[i] & 31u;
		touched_seq.push_back(seq[i]);
	}
	std::sort(touched_seq.begin(), touched_seq.end());
	touched_seq.erase(std::unique(touched_seq.begin(), touched_seq.end()), touched_seq.end());
	int rc;
	if(eos_changed) HIP_TRY(hipMemcpyAsync(S.d_has_eos.p, S.has_eos.data(), S.n, hipMemcpyHostToDevice, ctx->stream));
	// device: clear the bases in their blocks (one launch), refresh the 2 blocks of windows that can see each, redo the irregular
	// lists of the sequences concerned and upload the set's list ONCE (it is rebuilt whole: order by size counter, scan records --
	// tens of ms for C2's 1.2e6 words, which every one of a design iteration's ~70 splits paid when this took one split per call)
	if((rc = ctx->split_where.ensure(where.size())) != PCR_OK) return rc;
	HIP_TRY(hipMemcpyAsync(ctx->split_where.p, where.data(), where.size()*sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
	hipLaunchKernelGGL(k_apply_splits, dim3((n + 255)/256), dim3(256), 0, ctx->stream, ctx->split_where.p, n, (uint32_t *)S.planes.p, S.nib.p);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipStreamSynchronize(ctx->stream));                                   // `where` is a local
	for(uint32_t i = 0;i < n;++i){
		const uint64_t gb = where[2*(size_t)i];
		const uint64_t first = (gb > S.blk_off[seq[i]]) ? gb - 1 : gb;
		if((rc = run_valid(ctx, S, first, gb - first + 1)) != PCR_OK) return rc;
	}
	const auto t0 = std::chrono::steady_clock::now();
	for(uint32_t sq : touched_seq){
		S.irr_host[sq].clear();
		pcrhost::PackedSeq q; q.buf = S.packed[sq].data(); q.len = S.len[sq];
		if(!pcrhost::irregular_words(q, ctx->filt, S.irr_host[sq])){ g_err = "pcr_split: irregular word overflow"; return PCR_ERR_CAPACITY; }
	}
	S.have_db = false; S.have_codes = false;
	const auto t1 = std::chrono::steady_clock::now();
	rc = upload_irregular(ctx, S);
	if(ctx->timing) fprintf(stderr, "[pcramp] split_many: %u splits in %zu sequences; irregular words %.1f ms, list upload %.1f ms\n", n, touched_seq.size(),
		std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
	return rc;
}

int pcr_split(pcr_ctx *ctx, pcr_set which, uint32_t seq, uint64_t pos)
{
	return pcr_split_many(ctx, which, &seq, &pos, 1);
}

} // extern "C"

namespace {

inline uint32_t spread16(uint32_t v)               // bit i of the low half -> bit 2i
{
	v &= 0xFFFFu;
	v = (v | (v << 8)) & 0x00FF00FFu; v = (v | (v << 4)) & 0x0F0F0F0Fu;
	v = (v | (v << 2)) & 0x33333333u; return (v | (v << 1)) & 0x55555555u;
}

// The seeds of a pass for the second form of the seed scan: per orientation from the cache (derived on a miss), listed as
// code << 14 | slot offset << 9 | orientation WITHIN ITS GROUP, in groups of at most S2_MAX_OR whole orientations, each of which
// fits the LDS budget of one launch.  One launch per group: IUPAC primers expand to more 9-gram seeds than one launch holds
// (two launches of this form still beat the first form, 2 x ~65 vs 178 + 22 us at C5's shard), and the reference's default
// batch of 1 000 trial assays (pcramp.h:32) is 4 000 orientations = 16+ groups (r02 sent it to the first form with host-built
// tables: 21 ms per select_words on a C5 shard).  false: an orientation whose seeds alone exceed one launch.
constexpr uint32_t S2_MAX_GROUPS = 4096;
// S3 (optional): the set whose position index the third form will read -- the chunk list of every launch group (ctx->s3_prefix: the
// running number of 64-entry chunks of the group's seeds' runs, its total behind it) is made alongside, from chunk counts cached
// beside each oligo's seeds for the set last used.
bool plan_seed2(pcr_ctx *ctx, const std::vector<pcrhost::Candidate> &cand, std::vector<uint32_t> &or_seed, std::vector<uint32_t> &or_plain, uint32_t &irr_off_mask,
	const SeqSet *S3 = nullptr)
{
	std::vector<uint32_t> &pf = ctx->s3_prefix;
	pf.clear();
	uint32_t pf_run = 0;
	const uint32_t n_or = 2*(uint32_t)cand.size();
	std::vector<uint32_t> &out = ctx->s2_seeds;
	out.clear(); ctx->s2_group_end.clear(); ctx->s2_group_offmask.clear(); ctx->s2_group_or.clear(); ctx->s2_group_nor.clear();
	irr_off_mask = 0;
	ctx->s2_masks.resize(n_or);
	if(ctx->s2_cache.size() > 16384) ctx->s2_cache.clear();
	size_t group_begin = 0; uint32_t group_mask = 0, group_or0 = 0, group_last = 0;
	// the tables of a launch must fit one CU's LDS; the third form keeps only the masks and a slice of the lists there: more orientations
	// (9-bit ids) and as many seeds as the pass has -- plan_seed3_slices() checks its LDS need afterwards
	auto fits = [&](size_t n, uint32_t g_or){
		if(S3) return g_or <= S3_MAX_OR && n <= S3_MAX_SEEDS;
		return g_or <= S2_MAX_OR && n <= S2_MAX_SEEDS && sizeof(S2Shared) + 17*(size_t)g_or + 8*n + 1024 <= 160*1024;
	};
	for(uint32_t o = 0;o < n_or;++o){
		const pcrhost::Candidate &c = cand[o >> 1];
		const Planes &m = (o & 1u) ? c.rc : c.fwd;
		const pcr_ctx::S2Key key = {m.a, m.c, m.g, m.t, c.floor_};
		auto it = ctx->s2_cache.find(key);
		if(it == ctx->s2_cache.end()){
			pcr_ctx::S2Entry e; e.off_mask = 0;
			ctx->s2_tmp.clear();
			e.seedable = pcrhost::orientation_seeds(m, c.floor_, 0, ctx->s2_tmp, nullptr, S2_Q);
			// per orientation ONE 16-byte mask entry: the four base-set planes, slots 0..15 spread to the even bits (slot k -> bit 2k) and
			// slots 16..31 to the odd bits (slot 16 + k -> bit 2k + 1): both halves of a window are counted by one multiplexer pass
			// (s2_count) from one LDS read
			e.mask = make_uint4(spread16(m.a) | (spread16(m.a >> 16) << 1), spread16(m.c) | (spread16(m.c >> 16) << 1),
			                    spread16(m.g) | (spread16(m.g >> 16) << 1), spread16(m.t) | (spread16(m.t >> 16) << 1));
			e.seeds.reserve(ctx->s2_tmp.size());
			for(const pcrhost::Seed &sd : ctx->s2_tmp){ e.seeds.push_back((sd.code << 14) | ((uint32_t)sd.off << 9)); e.off_mask |= 1u << sd.off; }
			it = ctx->s2_cache.emplace(key, std::move(e)).first;
		}
		pcr_ctx::S2Entry &e = it->second;
		ctx->s2_masks[o] = e.mask;
		if(!e.seedable){ or_plain.push_back(o); continue; }
		or_seed.push_back(o);
		// the group spans orientations [group_or0, o]: unseedable ones in between only take an (unused) id
		if(!fits(out.size() - group_begin + e.seeds.size(), o - group_or0 + 1)){   // close the group, open the next at this orientation
			if(!fits(e.seeds.size(), 1) || ctx->s2_group_end.size() + 1 >= S2_MAX_GROUPS) return false;
			ctx->s2_group_end.push_back((uint32_t)out.size()); ctx->s2_group_offmask.push_back(group_mask); ctx->s2_group_or.push_back(group_or0); ctx->s2_group_nor.push_back(group_last - group_or0 + 1);
			group_begin = out.size(); group_mask = 0; group_or0 = o;
			if(S3){ pf.push_back(pf_run); pf_run = 0; }                                // the group's total; the next one starts at 0
		}
		group_last = o;
		const size_t at = out.size();
		out.resize(at + e.seeds.size());
		for(size_t k = 0;k < e.seeds.size();++k) out[at + k] = e.seeds[k] | (o - group_or0);
		if(S3){
			if(e.chunks_gen != S3->pix_generation){                                   // exclusive running chunk counts of the oligo's seeds in this set
				e.chunks.resize(e.seeds.size());
				uint32_t run = 0;
				for(size_t k = 0;k < e.seeds.size();++k){ e.chunks[k] = run; run += (S3->pix_count_h[e.seeds[k] >> 14] + 63u) >> 6; }
				e.chunks_total = run; e.chunks_gen = S3->pix_generation;
			}
			const size_t pa = pf.size();
			pf.resize(pa + e.chunks.size());
			for(size_t k = 0;k < e.chunks.size();++k) pf[pa + k] = pf_run + e.chunks[k];
			pf_run += e.chunks_total;
		}
		if(!(o & 1u)){ irr_off_mask |= e.off_mask; group_mask |= e.off_mask; }   // slot offsets at which forward seeds sit (irregular-word scan)
	}
	ctx->s2_group_end.push_back((uint32_t)out.size()); ctx->s2_group_offmask.push_back(group_mask); ctx->s2_group_or.push_back(group_or0);
	ctx->s2_group_nor.push_back(or_seed.empty() ? 0u : group_last - group_or0 + 1);
	if(S3) pf.push_back(pf_run);
	return true;
}

// Third form: the chunks of every launch group dealt to G workgroups in equal contiguous shares, and per workgroup the first seed of its
// share (a merge walk over the group's chunk list).  false: some group's largest slice does not fit the LDS (many seeds whose runs are
// empty side by side: tiny target sets) -- the caller plans again within the second form's limits.
constexpr size_t S3_LDS_BUDGET = 120*1024;
uint32_t seed3_grid(const pcr_ctx *ctx, int *wg_threads)
{
	static const int s3_wg = getenv("PCRAMP_S3_WG") ? atoi(getenv("PCRAMP_S3_WG")) : 1024;       // A/B: workgroup size, workgroups per CU
	static const int s3_per_cu = getenv("PCRAMP_S3_PER_CU") ? atoi(getenv("PCRAMP_S3_PER_CU")) : 0;
	if(wg_threads) *wg_threads = (s3_wg == 1024) ? 1024 : 512;
	return std::min<uint32_t>(ctx->n_cu*(uint32_t)(s3_per_cu ? s3_per_cu : (s3_wg == 1024 ? 2 : 4)), S3_MAX_WG);
}
bool plan_seed3_slices(pcr_ctx *ctx, bool with_irr)
{
	int wg_threads = 0;
	const uint32_t G_all = seed3_grid(ctx, &wg_threads);
	ctx->s3_launch.resize(ctx->s2_group_end.size());
	size_t g_begin = 0, g_prefix = 0;
	for(size_t g = 0;g < ctx->s2_group_end.size();++g){
		const uint32_t ns = ctx->s2_group_end[g] - (uint32_t)g_begin;
		g_begin = ctx->s2_group_end[g];
		const uint32_t *P = ctx->s3_prefix.data() + g_prefix;                  // the group's chunk list, [ns + 1]
		g_prefix += (size_t)ns + 1;
		pcr_ctx::S3Launch &L = ctx->s3_launch[g];
		// the launch's first workgroups look the irregular words up (16 seeds per wave turn) and leave the chunks to the others: the two
		// chains of dependent loads then run side by side; at most a quarter of the launch
		L.n_irr_wg = with_irr ? std::min<uint32_t>((ns + 16u*(uint32_t)(wg_threads/64) - 1u)/(16u*(uint32_t)(wg_threads/64)), G_all/4u) : 0u;
		const uint32_t G = G_all - L.n_irr_wg;
		L.n_chunks = P[ns]; L.per_wg = std::max<uint32_t>(1u, (L.n_chunks + G - 1u)/G); L.slice_cap = 2;
		if(ns == 0) continue;
		uint32_t at = 0;
		for(uint32_t w = 0;w <= G;++w){
			const uint64_t target = (uint64_t)w*L.per_wg;
			while(at + 1u < ns && P[at + 1u] <= target) ++at;
			L.W.start[w] = at;
		}
		for(uint32_t w = G + 1;w <= S3_MAX_WG;++w) L.W.start[w] = at;
		for(uint32_t w = 0;w < G;++w) L.slice_cap = std::max(L.slice_cap, std::min(L.W.start[w + 1] + 2u, ns + 1u) - L.W.start[w]);
		if(17*(size_t)ctx->s2_group_nor[g] + 8*(size_t)L.slice_cap + 64 > S3_LDS_BUDGET) return false;
	}
	return true;
}

// The seeds of a pass for the first form with device-built tables: per orientation from the cache (8-gram seeds), listed as
// code | slot offset << 16 | orientation << 21.  When the lists exceed S1_MAX_SEEDS (low thresholds: hundreds of codes per
// orientation) the orientations with the longest lists go to the bit-sliced scan.
void plan_seed1(pcr_ctx *ctx, const std::vector<pcrhost::Candidate> &cand, std::vector<uint32_t> &or_seed, std::vector<uint32_t> &or_plain, uint32_t &irr_off_mask)
{
	const uint32_t n_or = 2*(uint32_t)cand.size();
	std::vector<uint32_t> &out = ctx->s1_seeds;
	out.clear();
	irr_off_mask = 0;
	if(ctx->s1_cache.size() > 16384) ctx->s1_cache.clear();
	std::vector<const pcr_ctx::S2Entry *> ent(n_or);
	size_t total = 0;
	for(uint32_t o = 0;o < n_or;++o){
		const pcrhost::Candidate &c = cand[o >> 1];
		const Planes &m = (o & 1u) ? c.rc : c.fwd;
		const pcr_ctx::S2Key key = {m.a, m.c, m.g, m.t, c.floor_};
		auto it = ctx->s1_cache.find(key);
		if(it == ctx->s1_cache.end()){
			pcr_ctx::S2Entry e; e.off_mask = 0;
			ctx->s2_tmp.clear();
			e.seedable = pcrhost::orientation_seeds(m, c.floor_, 0, ctx->s2_tmp);
			e.seeds.reserve(ctx->s2_tmp.size());
			for(const pcrhost::Seed &sd : ctx->s2_tmp){ e.seeds.push_back(sd.code | ((uint32_t)sd.off << 16)); e.off_mask |= 1u << sd.off; }
			it = ctx->s1_cache.emplace(key, std::move(e)).first;
		}
		ent[o] = &it->second;                                            // (unordered_map: references stay valid across insertions)
		if(ent[o]->seedable) total += ent[o]->seeds.size();
	}
	std::vector<uint8_t> drop(n_or, 0);
	if(total > S1_MAX_SEEDS){
		std::vector<uint32_t> by_len;
		for(uint32_t o = 0;o < n_or;++o){ if(ent[o]->seedable) by_len.push_back(o); }
		std::stable_sort(by_len.begin(), by_len.end(), [&](uint32_t x, uint32_t y){ return ent[x]->seeds.size() > ent[y]->seeds.size(); });
		for(size_t i = 0;i < by_len.size() && total > S1_MAX_SEEDS;++i){ drop[by_len[i]] = 1; total -= ent[by_len[i]]->seeds.size(); }
	}
	out.resize(total);
	size_t at = 0;
	for(uint32_t o = 0;o < n_or;++o){
		const pcr_ctx::S2Entry &e = *ent[o];
		if(!e.seedable || drop[o]){ or_plain.push_back(o); continue; }
		or_seed.push_back(o);
		const uint32_t tag = o << 21;
		for(size_t k = 0;k < e.seeds.size();++k) out[at + k] = e.seeds[k] | tag;
		at += e.seeds.size();
		if(!(o & 1u)) irr_off_mask |= e.off_mask;                       // slot offsets at which forward seeds sit (irregular-word scan)
	}
}

// The first form's plan when its tables are built on the HOST (passes with 5'/3' shift candidates, more than S1_MAX_OR
// orientations, PCRAMP_SEED_TABLES=host): seeds per orientation (shift candidates inherit the unshifted oligo's), then the
// tables of pcr_scan_seed.inc: presence bitmap + rank (the LDS image), one head word per distinct code, grouped seed lists
// of the codes shared by several seeds.  Too many distinct codes, or a code shared by more than 255 seeds, sends the densest
// quarter of the orientations (with shift candidates: everything) to the bit-sliced path.
struct HostSeedPlan {
	std::vector<pcrhost::Seed> seeds;
	std::vector<std::vector<std::pair<uint16_t, int8_t> > > inheritors;   // per orientation: (shifted orientation, shift) sharing its seeds
	size_t n_inherited = 0;
	std::vector<uint32_t> image, heads, multi;
};

int plan_seed_host(pcr_ctx *ctx, const std::vector<pcrhost::Candidate> &cand, HostSeedPlan &H, std::vector<uint32_t> &or_seed, std::vector<uint32_t> &or_plain,
	uint32_t &irr_off_mask)
{
	const uint32_t n_or = 2*(uint32_t)cand.size();
	struct OrientInfo { uint32_t begin, end; int max_exact_pos; bool seeded; };
	std::vector<OrientInfo> info;
	std::vector<pcrhost::Seed> &seeds = H.seeds;
	std::vector<std::vector<std::pair<uint16_t, int8_t> > > &inheritors = H.inheritors;
	size_t &n_inherited = H.n_inherited;
	std::vector<uint32_t> &image = H.image, &heads = H.heads, &multi = H.multi;
	// a 5'/3' shift candidate inherits the seeds of the unshifted oligo, moved by its shift, as long as no
	// padded 8-window would have to be clamped at the end of the word (it costs 1/10 of deriving them anew)
	info.assign(n_or, OrientInfo());
	inheritors.assign(n_or, std::vector<std::pair<uint16_t, int8_t> >());
	seeds.reserve((size_t)n_or*32);
	for(uint32_t o = 0;o < n_or;++o){
		const pcrhost::Candidate &c = cand[o >> 1];
		OrientInfo &me = info[o];
		me.begin = (uint32_t)seeds.size(); me.max_exact_pos = -1;
		const uint32_t bo = 2*c.base + (o & 1u);
		const int32_t sh = (o & 1u) ? -c.shift : c.shift;
		if(c.base != (o >> 1) && info[bo].seeded && info[bo].max_exact_pos <= 24 && info[bo].max_exact_pos + sh <= 24){
			inheritors[bo].push_back(std::make_pair((uint16_t)o, (int8_t)sh));   // its seeds = those of bo with off + sh: expanded when the table is built
			n_inherited += info[bo].end - info[bo].begin;
			me.seeded = true; me.max_exact_pos = (info[bo].max_exact_pos < 0) ? -1 : info[bo].max_exact_pos + sh;
		}
		else me.seeded = pcrhost::orientation_seeds((o & 1) ? c.rc : c.fwd, c.floor_, o, seeds, &me.max_exact_pos);
		me.end = (uint32_t)seeds.size();
		if(me.seeded) or_seed.push_back(o); else or_plain.push_back(o);
	}
	if(or_seed.empty()) return PCR_OK;
	// count[] / own[] (one entry per 8-gram code) are kept all-zero between passes: only the entries a pass touched
	// are cleared again (two 64K-entry memsets per pass were ~8 us of the host plan)
	if(ctx->seed_count.size() != 65536){ ctx->seed_count.assign(65536, 0); ctx->seed_own.assign(65536, 0); }
	std::vector<uint16_t> &count = ctx->seed_count;
	bool overflow = false;
	uint32_t distinct = 0;
	std::vector<uint8_t> &own = ctx->seed_own;                          // seeds listed under the code (its inheritors come on top)
	for(;;){
		image.assign(SEED_IMAGE_WORDS, 0u);
		overflow = false; distinct = 0;
		for(const pcrhost::Seed &sd : seeds){
			uint16_t &c = count[sd.code];
			if(c == 0){ image[sd.code >> 5] |= 1u << (sd.code & 31); ++distinct; }
			c = (uint16_t)(c + 1 + (inheritors.empty() ? 0 : inheritors[sd.orient].size()));
			if(c > 255){ overflow = true; break; }
			++own[sd.code];
		}
		if(distinct > SEED_MAX_DISTINCT) overflow = true;
		if(!overflow || n_inherited || or_seed.size() < 2) break;
		// Too dense (low thresholds: hundreds of codes per orientation): hand the quarter of the seeded orientations
		// with the longest code lists to the bit-sliced scan and count again.
		for(const pcrhost::Seed &sd : seeds){ count[sd.code] = 0; own[sd.code] = 0; }
		std::vector<uint32_t> by_len(or_seed);
		std::stable_sort(by_len.begin(), by_len.end(), [&](uint32_t x, uint32_t y){ return info[x].end - info[x].begin > info[y].end - info[y].begin; });
		const size_t n_drop = (by_len.size() + 3)/4;
		std::vector<uint8_t> drop(n_or, 0);
		for(size_t i = 0;i < n_drop;++i) drop[by_len[i]] = 1;
		std::vector<pcrhost::Seed> kept; kept.reserve(seeds.size());
		for(uint32_t o = 0;o < n_or;++o){
			const uint32_t b = info[o].begin, e = info[o].end;
			info[o].begin = (uint32_t)kept.size();
			if(!drop[o]) kept.insert(kept.end(), seeds.begin() + b, seeds.begin() + e);
			info[o].end = (uint32_t)kept.size();
			if(drop[o]) info[o].seeded = false;
		}
		seeds.swap(kept);
		or_seed.clear(); or_plain.clear();
		for(uint32_t o = 0;o < n_or;++o){ if(info[o].seeded) or_seed.push_back(o); else or_plain.push_back(o); }
	}
	for(const pcrhost::Seed &sd : seeds){                                 // slot offsets at which forward seeds sit (irregular-word scan)
		if(sd.orient & 1u) continue;
		irr_off_mask |= 1u << sd.off;
		if(!inheritors.empty()){ for(const std::pair<uint16_t, int8_t> &in : inheritors[sd.orient]) irr_off_mask |= 1u << (sd.off + in.second); }
	}
	if(overflow){
		or_plain.clear(); or_seed.clear(); image.clear();
		for(uint32_t o = 0;o < n_or;++o) or_plain.push_back(o);
	}
	else{
		uint16_t *rank16 = (uint16_t *)(image.data() + SEED_BITMAP_WORDS);
		uint32_t run = 0;
		for(uint32_t w = 0;w < SEED_BITMAP_WORDS;++w){ rank16[w] = (uint16_t)run; run += (uint32_t)__builtin_popcount(image[w]); }
		heads.assign(distinct, 0u);
		// first pass: reserve the multi ranges (head = start << 8 | filled so far); second: fill
		uint32_t n_multi = 0;
		for(const pcrhost::Seed &sd : seeds){
			const uint32_t h = rank16[sd.code >> 5] + (uint32_t)__builtin_popcount(image[sd.code >> 5] & ((1u << (sd.code & 31)) - 1u));
			if(count[sd.code] == 1){ heads[h] = SEED_SINGLE | ((uint32_t)sd.orient << 8) | sd.off; continue; }
			if(heads[h] == 0){ heads[h] = 0x40000000u | n_multi; n_multi += own[sd.code]; }   // bit 30: range reserved, low bits = start
		}
		std::vector<uint32_t> raw(n_multi, 0u);                  // per code, in generation order: orient | off << 24
		std::vector<uint8_t> &fill = ctx->seed_fill; fill.assign(distinct, 0);
		for(const pcrhost::Seed &sd : seeds){
			if(count[sd.code] == 1) continue;
			const uint32_t h = rank16[sd.code >> 5] + (uint32_t)__builtin_popcount(image[sd.code >> 5] & ((1u << (sd.code & 31)) - 1u));
			const uint32_t start = heads[h] & 0x3FFFFFFFu;
			raw[start + fill[h]++] = (uint32_t)sd.orient | ((uint32_t)sd.off << 24);
		}
		// Group the seeds of a code: slot shifts of one oligo orientation (consecutive candidates, so consecutive
		// here) whose base-aligned offset boff = off - shift is the same lie over the same target bases; their
		// match count is taken once, with the unshifted orientation at window x - boff (layout: pcr_scan_seed.inc).
		multi.clear(); multi.reserve(n_multi + n_multi/2 + 16);
		for(uint32_t h = 0;h < distinct;++h){
			if(heads[h] & SEED_SINGLE) continue;
			const uint32_t start = heads[h] & 0x3FFFFFFFu, n = fill[h];
			const uint32_t out0 = (uint32_t)multi.size();
			uint32_t header_at = 0, members = 0, cur_base = 0xFFFFFFFFu; int32_t cur_boff = -1;
			for(uint32_t e = 0;e < n;++e){
				const uint32_t sd = raw[start + e], orient = sd & 0xFFFFu, off = sd >> 24;
				const pcrhost::Candidate &c = cand[orient >> 1];
				const int32_t sh = (orient & 1u) ? -c.shift : c.shift;
				int32_t boff = (int32_t)off - sh;
				uint32_t base_orient = 2*c.base + (orient & 1u);
				if(boff < 0 || boff > 24){ boff = (int32_t)off; base_orient = orient; }   // the unshifted window would leave the word: stands alone
				if(members && base_orient == cur_base && boff == cur_boff && members < 256){
					multi.push_back(sd); ++members;
					multi[header_at] = SEED_GROUP | cur_base | ((uint32_t)cur_boff << 16) | ((members - 1) << 21);
				}
				else{
					header_at = (uint32_t)multi.size(); members = 1; cur_base = base_orient; cur_boff = boff;
					multi.push_back(SEED_GROUP | cur_base | ((uint32_t)cur_boff << 16));
					multi.push_back(sd);
				}
				// the shift candidates that inherit this seed: same target bases, off moved by their shift
				if(!inheritors.empty()){
					for(const std::pair<uint16_t, int8_t> &in : inheritors[orient]){
						if(members == 256){        // header field full: open another group with the same leader
							header_at = (uint32_t)multi.size(); members = 0;
							multi.push_back(SEED_GROUP | cur_base | ((uint32_t)cur_boff << 16));
						}
						multi.push_back((uint32_t)in.first | ((uint32_t)((int32_t)off + in.second) << 24)); ++members;
						multi[header_at] = SEED_GROUP | cur_base | ((uint32_t)cur_boff << 16) | ((members - 1) << 21);
					}
				}
			}
			heads[h] = (out0 << 9) | ((uint32_t)multi.size() - out0);
		}
		n_multi = (uint32_t)multi.size();
		if(n_multi >= (1u << 22)){ ctx->seed_count.clear(); g_err = "pcr_select_words: seed table too large"; return PCR_ERR_CAPACITY; }
	}
	return PCR_OK;
}

// pcr_select_words proper.  async: enqueue one attempt and return without looking at the counters
// (pcr_screen_device; the caller records the pass as pending).
int select_impl(pcr_ctx *ctx, pcr_set which, const pcr_pair *pairs, uint32_t n_pairs, int optimize_5, int optimize_3,
	float threshold, uint32_t min_oligo_length, uint64_t *n_entries_out, bool async, FusedAmp *fa = nullptr)
{
	if(!ctx || (n_pairs && !pairs)){ g_err = "pcr_select_words: bad argument"; return PCR_ERR_ARG; }
	if(min_oligo_length < 1 || min_oligo_length > 32){ g_err = "pcr_select_words: min_oligo_length must be in [1,32]"; return PCR_ERR_ARG; }
	HIP_TRY(hipSetDevice(ctx->device));
	SeqSet &S = ctx->sets[which];
	S.have_db = false; S.n_entries = 0;
	if(n_entries_out) *n_entries_out = 0;
	const bool ctrl_was_clean = S.ctrl_clean;         // left so by the fused tail of the previous pass over this set
	S.ctrl_clean = false; S.touched_from_seg = false;
	bool lean = false, cleared_bits = false; size_t lean_bits_bytes = 0;   // lean: the fused pass without a staging launch (see pcr_ctx::dstage)
	HostTimer timer(ctx, 0);
	if(ctx->timing) ++ctx->n_timed;
	std::vector<pcrhost::Candidate> cand;
	pcrhost::build_candidates((const uint64_t *)pairs, n_pairs, optimize_5 != 0, optimize_3 != 0, threshold, cand);
	const uint32_t ncand = (uint32_t)cand.size();
	if(S.n == 0 || ncand == 0){ S.have_db = true; S.n_touched = 0; return PCR_OK; }
	std::vector<uint4> hf(ncand), hr(ncand); std::vector<uint32_t> hfl(ncand);
	for(uint32_t c = 0;c < ncand;++c){
		hf[c] = make_uint4(cand[c].fwd.a, cand[c].fwd.c, cand[c].fwd.g, cand[c].fwd.t);
		hr[c] = make_uint4(cand[c].rc.a, cand[c].rc.c, cand[c].rc.g, cand[c].rc.t);
		hfl[c] = cand[c].floor_;
	}
	int rc;
	if((rc = ctx->best.ensure((size_t)S.n*ncand)) != PCR_OK) return rc;
	if(ctx->best.generation != ctx->best_seen){
		// fresh (uninitialised) storage: clear once; afterwards the epoch tag makes clearing unnecessary
		HIP_TRY(hipMemsetAsync(ctx->best.p, 0, ctx->best.cap*sizeof(uint32_t), ctx->stream));
		ctx->best_seen = ctx->best.generation; ctx->epoch = std::min(ctx->debug_epoch, EPOCH_LIMIT);
	}
	// ---- scan plan.  version 3 (default): orientations that can be seeded go through the pigeonhole seed
	// scan; the others, and every tile holding IUPAC target codes, through the bit-sliced counter.
	// version 2: bit-sliced counter for everything.  version 1: one popcount per (window, orientation).
	const uint32_t n_or = 2*ncand;
	std::vector<uint32_t> or_seed, or_plain;          // orientation ids
	HostSeedPlan H;                                   // first form with host-built tables only
	uint32_t irr_off_mask = 0;
	// The second form of the seed scan (pcr_scan_seed2.inc) takes the pass when no 5'/3' shift candidates are asked for and
	// the orientations and their 9-gram seeds fit its LDS budget; the host then only LISTS the seeds (from a cache keyed by
	// oligo and floor: between two optimiser iterations most oligos stay what they were).
	bool want_seed3 = false, s3_with_irr = false;       // (the latter: the set has live irregular words, which the third form looks up through their index)
	if(ctx->scan_version == 3 && !ctx->force_seed1 && !optimize_5 && !optimize_3 && n_or <= 65535 && !ctx->no_seed3 && !ctx->no_irr_index && ctx->s2_dbg == 0){
		// the position index costs 16 bytes per base and milliseconds to build: a set gets it when the first pass that can use it
		// arrives (every candidate seeded) -- a background set screened at 0.72 never does
		bool worth = true;
		if(!S.pix_valid){
			std::vector<uint32_t> os, op; uint32_t om = 0;
			worth = plan_seed2(ctx, cand, os, op, om, nullptr) && op.empty() && !os.empty();
		}
		if(worth){
			int irc = ensure_pos_index(ctx, S);
			if(irc != PCR_OK) return irc;
			uint32_t n_live0 = 0;
			for(uint32_t k = std::min<uint32_t>(min_oligo_length, 256);k < 256;++k) n_live0 += S.irr_size_count[k];
			bool irr_ok = n_live0 == 0;
			if(S.pix_usable && !irr_ok && S.irr_n_multi == 0 && (uint64_t)24*S.n_irr < (uint64_t(1) << 32)){
				if((irc = ensure_irr_index(ctx, S)) != PCR_OK) return irc;
				irr_ok = S.irx_usable;
			}
			want_seed3 = S.pix_usable && irr_ok; s3_with_irr = n_live0 > 0;
		}
	}
	bool use_seed2 = false;
	if(ctx->scan_version == 3 && !ctx->force_seed1 && !optimize_5 && !optimize_3 && n_or <= 65535){
		use_seed2 = plan_seed2(ctx, cand, or_seed, or_plain, irr_off_mask, want_seed3 ? &S : nullptr);
		if(want_seed3 && !(use_seed2 && or_plain.empty() && !or_seed.empty() && plan_seed3_slices(ctx, s3_with_irr))){
			// the third form will not take the pass (an unseeded candidate, or its lists do not fit): plan within the second form's limits
			or_seed.clear(); or_plain.clear();
			use_seed2 = plan_seed2(ctx, cand, or_seed, or_plain, irr_off_mask, nullptr);
		}
		// an orientation without a 9-gram structure (low thresholds: k = 4 mismatching slots and more) may still have an 8-gram
		// one: let the first form plan the pass where it can (it hands fewer orientations to the bit-sliced scan); a batch beyond
		// its S1_MAX_OR orientations keeps this form for the seedable orientations, the others go to the bit-sliced scan
		if(use_seed2 && !or_plain.empty() && (n_or <= S1_MAX_OR || or_seed.empty())) use_seed2 = false;
		if(!use_seed2){ or_seed.clear(); or_plain.clear(); irr_off_mask = 0; }
	}
	// ... and the third form -- the targets' positions indexed by their 9-grams, the seeds looked up (pcr_scan_seed3.inc) -- where every
	// candidate is seeded (and the irregular words can come in through their index too: want_seed3, decided before the planning)
	const bool use_seed3 = use_seed2 && want_seed3 && or_plain.empty() && !or_seed.empty()
		&& ctx->s3_prefix.size() == ctx->s2_seeds.size() + ctx->s2_group_end.size();
	bool dev_tables = false;                           // first form, tables built by k_seed_tables
	if(!use_seed2 && ctx->scan_version == 3 && !ctx->host_seed_tables && !optimize_5 && !optimize_3 && n_or <= S1_MAX_OR){
		plan_seed1(ctx, cand, or_seed, or_plain, irr_off_mask);
		dev_tables = true;
	}
	if(use_seed2 || dev_tables){ /* planned */ }
	else if(ctx->scan_version == 3 && n_or <= 65535){
		const int prc = plan_seed_host(ctx, cand, H, or_seed, or_plain, irr_off_mask);
		// (the counters the table build touched are cleared again whatever happened: they stay all-zero between passes)
		if(ctx->seed_count.size() == 65536){ for(const pcrhost::Seed &sd : H.seeds){ ctx->seed_count[sd.code] = 0; ctx->seed_own[sd.code] = 0; } }
		if(prc != PCR_OK) return prc;
	}
	else{ for(uint32_t o = 0;o < n_or;++o) or_plain.push_back(o); }
	SeedTables ST; memset(&ST, 0, sizeof(ST));
	Seed2Tables ST2; memset(&ST2, 0, sizeof(ST2));
	const uint32_t *d_s3_prefix = nullptr;
	const size_t n_seeds = use_seed2 ? ctx->s2_seeds.size() : dev_tables ? ctx->s1_seeds.size() : H.seeds.size() + H.n_inherited;
	if(ctx->debug_log) fprintf(stderr, "[pcramp] scan plan: %u candidates, %zu seeded orientations (%zu seeds), %zu plain, %u/%u IUPAC tiles, %u-slot buckets\n",
		ncand, or_seed.size(), n_seeds, or_plain.size(), S.n_degen_tiles, S.n_tiles, S.bucket_cap);
	Scan2Tables tab_plain, tab_seedset;               // bit-sliced tables: unseedable orientations (all tiles) / seedable ones (IUPAC tiles)
	const bool need_plain = (ctx->scan_version != 1) && !or_plain.empty();
	const bool need_seedset = !or_seed.empty() && S.n_degen_tiles > 0;
	if(need_plain) build_scan2_tables(cand, or_plain, tab_plain);
	if(need_seedset) build_scan2_tables(cand, or_seed, tab_seedset);
	const uint32_t *d_tab_plain = nullptr, *d_bias_plain = nullptr, *d_map_plain = nullptr;
	const uint32_t *d_tab_seedset = nullptr, *d_bias_seedset = nullptr, *d_map_seedset = nullptr;
	{
		size_t bytes = ncand*(2*sizeof(uint4) + sizeof(uint32_t)) + 1024;
		bytes += (tab_plain.tab.size() + tab_plain.bias.size() + or_plain.size() + 256)*sizeof(uint32_t);
		bytes += (tab_seedset.tab.size() + tab_seedset.bias.size() + or_seed.size() + 256)*sizeof(uint32_t);
		bytes += (H.image.size() + H.heads.size() + H.multi.size() + 64)*sizeof(uint32_t);
		std::vector<uint4> &masks2 = ctx->s2_masks; std::vector<uint8_t> &floors2 = ctx->s2_floors;
		if(use_seed2){
			// (the mask entries come out of the per-oligo cache: plan_seed2)
			floors2.assign(((size_t)n_or + 15) & ~size_t(15), 0);
			for(uint32_t o = 0;o < n_or;++o) floors2[o] = (uint8_t)std::min<uint32_t>(cand[o >> 1].floor_, 255u);
			bytes += masks2.size()*sizeof(uint4) + floors2.size() + ctx->s2_seeds.size()*sizeof(uint32_t) + 512;
			if(use_seed3) bytes += ctx->s3_prefix.size()*sizeof(uint32_t) + 64;
		}
		const bool build_tables = dev_tables && !or_seed.empty();
		if(build_tables) bytes += ctx->s1_seeds.size()*sizeof(uint32_t) + 256;
		// fused pass: the amplicon screen's oligo table travels with the scan tables, its result bitsets are
		// cleared by the same launch
		std::vector<OligoDev> ol;
		size_t bits_bytes = 0;
		bool fuse = false;
		if(fa && fa->n_pairs){
			bits_bytes = (size_t)fa->n_pairs*((S.n + 63)/64)*sizeof(uint64_t);
			fuse = ((uintptr_t)fa->d_fr % 16 == 0) && ((uintptr_t)fa->d_rf % 16 == 0) && (bits_bytes % 16 == 0);
			if(fuse){ build_oligos(fa->pairs, fa->n_pairs, fa->a, ol); bytes += ol.size()*sizeof(OligoDev) + 64; }
		}
		timer.next(1);
		// the pass's control block (counters | per-sequence fills | segment ends)
		{
			const uint64_t gen = S.ctrl.generation;
			if((rc = S.ctrl.ensure(8 + 2*(size_t)S.n + 4)) != PCR_OK) return rc;
			lean = ctx->direct_ok && async && fuse && use_seed2 && !or_seed.empty() && ctrl_was_clean && gen == S.ctrl.generation
				&& S.bucket_cap == POST_CAP && 2*fa->n_pairs <= 32*POST_MASK_WORDS;
			lean_bits_bytes = bits_bytes;
		}
		if(ctx->debug_log) fprintf(stderr, "[pcramp] staging: %s\n", lean ? "lean (tables written into device memory, no staging launch)" : "k_stage");
		Stager st(ctx);
		if((rc = lean ? st.begin_direct(bytes) : st.begin(bytes)) != PCR_OK) return rc;
		ctx->d_cand_fwd = st.put(hf.data(), ncand);
		ctx->d_cand_rc = st.put(hr.data(), ncand);
		ctx->d_cand_floor = st.put(hfl.data(), ncand);
		auto pad256 = [](std::vector<uint32_t> v){ v.resize((v.size() + 255) & ~size_t(255), 0xFFFFFFFFu); return v; };
		if(need_plain){
			d_tab_plain = st.put(tab_plain.tab.data(), tab_plain.tab.size());
			d_bias_plain = st.put(tab_plain.bias.data(), tab_plain.bias.size());
			const std::vector<uint32_t> m = pad256(or_plain);
			d_map_plain = st.put(m.data(), m.size());
		}
		if(need_seedset){
			d_tab_seedset = st.put(tab_seedset.tab.data(), tab_seedset.tab.size());
			d_bias_seedset = st.put(tab_seedset.bias.data(), tab_seedset.bias.size());
			const std::vector<uint32_t> m = pad256(or_seed);
			d_map_seedset = st.put(m.data(), m.size());
		}
		if(!H.image.empty()){
			ST.image = st.put(H.image.data(), H.image.size());
			ST.heads = st.put(H.heads.data(), H.heads.size());
			ST.multi = H.multi.empty() ? ST.heads : st.put(H.multi.data(), H.multi.size());
		}
		const uint32_t *d_s1_seeds = nullptr;
		if(build_tables){
			d_s1_seeds = st.put(ctx->s1_seeds.data(), ctx->s1_seeds.size());
			if((rc = ctx->s1_image.ensure(SEED_IMAGE_WORDS)) != PCR_OK) return rc;
			if((rc = ctx->s1_heads.ensure(2*(size_t)S1_MAX_SEEDS)) != PCR_OK) return rc;   // two words per distinct code
			if((rc = ctx->s1_multi.ensure(S1_MAX_SEEDS)) != PCR_OK) return rc;
			ST.image = ctx->s1_image.p; ST.heads = ctx->s1_heads.p; ST.multi = ctx->s1_multi.p; ST.flat = 1;
		}
		if(use_seed2){
			ST2.seeds = st.put(ctx->s2_seeds.data(), ctx->s2_seeds.size());
			ST2.masks = st.put(masks2.data(), masks2.size());
			ST2.floors = st.put(floors2.data(), floors2.size());
			ST2.n_seeds = (uint32_t)ctx->s2_seeds.size(); ST2.n_or = n_or;
			if(use_seed3) d_s3_prefix = st.put(ctx->s3_prefix.data(), ctx->s3_prefix.size());
		}
		// the staging launch also clears the control block and the result bitsets -- unless the pass is lean: then the tables
		// are already in device memory, the control block was left clean by the previous pass's tail and k_seed2 clears the bitsets
		if(fuse){
			fa->d_oligos = st.put(ol.data(), ol.size());
			fa->staged = true;
			if(lean) st.seal(ctx->mail_seq + 1);
			else if((rc = st.ship(S.ctrl.p, (8 + 2*(size_t)S.n)*sizeof(uint32_t), fa->d_fr, bits_bytes, fa->d_rf, bits_bytes, ctx->mail_seq + 1)) != PCR_OK) return rc;
		}
		else if((rc = st.ship(S.ctrl.p, (8 + 2*(size_t)S.n)*sizeof(uint32_t), nullptr, 0, nullptr, 0, ctx->mail_seq + 1)) != PCR_OK) return rc;
		if(build_tables){
			if((rc = ctx->s1_part.ensure(2*S1_GROUPS)) != PCR_OK) return rc;
			hipLaunchKernelGGL(k_seed_tables<false>, dim3(S1_GROUPS), dim3(S1_BUILD_THREADS), 0, ctx->stream, d_s1_seeds, (uint32_t)ctx->s1_seeds.size(),
				ctx->s1_part.p, ctx->s1_image.p, ctx->s1_heads.p, ctx->s1_multi.p);
			hipLaunchKernelGGL(k_seed_tables<true>, dim3(S1_GROUPS), dim3(S1_BUILD_THREADS), 0, ctx->stream, d_s1_seeds, (uint32_t)ctx->s1_seeds.size(),
				ctx->s1_part.p, ctx->s1_image.p, ctx->s1_heads.p, ctx->s1_multi.p);
			HIP_TRY(hipGetLastError());
		}
	}

	timer.next(2);
	uint32_t h_counters[4];
	if((rc = S.touched.ensure(S.n)) != PCR_OK) return rc;
	uint32_t *const d_counters = S.ctrl.p, *const d_seq_count = S.ctrl.p + 8;
	S.d_seg_hi = S.ctrl.p + 8 + S.n;
	for(int attempt = 0;;++attempt){
		const uint32_t cap = S.bucket_cap;
		const uint64_t n_slots = (uint64_t)S.n*cap;
		if(n_slots >= (uint64_t(1) << 32) || n_slots*(sizeof(Hit) + sizeof(DevEntry)) > (uint64_t(96) << 30)){
			g_err = "pcr_select_words: the per-sequence hit buckets would not fit (too many tied sites per sequence)"; return PCR_ERR_CAPACITY;
		}
		if((rc = ctx->hits.ensure(n_slots)) != PCR_OK) return rc;
		if((rc = S.db.ensure(n_slots)) != PCR_OK) return rc;
		if(attempt > 0) HIP_TRY(hipMemsetAsync(S.ctrl.p, 0, (8 + 2*(size_t)S.n)*sizeof(uint32_t), ctx->stream));
		if(ctx->epoch >= EPOCH_LIMIT){
			// the tag (epoch << 8 | count) is about to leave its 24 bits: every stale entry would compare HIGHER than the new
			// pass's hits and swallow them.  Checked per attempt (every bucket-growth retry takes an epoch of its own).
			HIP_TRY(hipMemsetAsync(ctx->best.p, 0, ctx->best.cap*sizeof(uint32_t), ctx->stream));
			ctx->epoch = 0;
		}
		++ctx->epoch;
		HitSink sink; sink.best = ctx->best.p; sink.hits = ctx->hits.p; sink.seq_count = d_seq_count;
		sink.counters = d_counters; sink.cap = cap; sink.ncand = ncand; sink.epoch = ctx->epoch;
		uint32_t n_live = 0;                              // irregular words whose size counter reaches min_oligo_length
		for(uint32_t k = std::min<uint32_t>(min_oligo_length, 256);k < 256;++k) n_live += S.irr_size_count[k];
		bool irr_fused = false;                           // scanned by extra workgroups of k_seed
		if(S.n_tiles){
			// events around the scan launches of every prof_stride-th pass (an event between two kernels costs a ~6 us queue bubble)
			ProfScope scan_prof(ctx, PCR_PROF_SCAN, ctx->prof && (ctx->prof_pass++ % ctx->prof_stride) == 0);
			if(ctx->scan_version == 1){
				hipLaunchKernelGGL(k_scan, dim3(S.n_tiles), dim3(SCAN_THREADS), 0, ctx->stream, S.planes.p, S.valid_d(),
					S.d_blk_off.p, S.d_len.p, S.d_active.p, S.tile_seq.p, S.tile_pos0.p, ctx->d_cand_fwd, ctx->d_cand_rc,
					ctx->d_cand_floor, ncand, sink);
				HIP_TRY(hipGetLastError());
			}
			else{
				if(need_plain && (rc = launch_scan2(ctx, S, tab_plain, ncand, sink, d_tab_plain, d_bias_plain, nullptr, S.n_tiles, d_map_plain)) != PCR_OK) return rc;
				if(!or_seed.empty() && use_seed2){
					// persistent workgroups of 16 waves, one per CU (the tables they build take most of its LDS); the irregular words
					// are taken by the same waves once their tiles are done.  One launch per seed group (plan_seed2).
					const uint32_t tiles_per_wg = S2_WAVES*2;
					const dim3 sgrid(std::max<uint32_t>(1u, std::min<uint32_t>((S.n_tiles + tiles_per_wg - 1)/tiles_per_wg, ctx->n_cu))), sblock(S2_THREADS);
					irr_fused = true;
					if(!ctx->s2_attr_set){
						HIP_TRY(hipFuncSetAttribute((const void *)k_seed2<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160*1024 - sizeof(S2Shared))));
						HIP_TRY(hipFuncSetAttribute((const void *)k_seed2<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160*1024 - sizeof(S2Shared))));
						ctx->s2_attr_set = true;
					}
					uint32_t g_begin = 0, g_prefix = 0;
					bool first_launch = true;
					bool irr_by_index = or_plain.empty() && S.irr_n_multi == 0 && n_live > 0 && !ctx->no_irr_index && (uint64_t)24*S.n_irr < (uint64_t(1) << 32);   // (the index counts its entries in 32 bits)
					for(size_t g = 0, b = 0;g < ctx->s2_group_end.size();++g){ if((size_t)ctx->s2_group_end[g] - b > (size_t)sgrid.x*S2_THREADS) irr_by_index = false; b = ctx->s2_group_end[g]; }   // (a thread looks up at most one seed)
					if(irr_by_index){ if((rc = ensure_irr_index(ctx, S)) != PCR_OK) return rc; irr_by_index = S.irx_usable; }
					if(use_seed3) irr_by_index = n_live > 0;                                 // (decided with the form: the index is there and usable)
					for(size_t g = 0;g < ctx->s2_group_end.size();++g){
						Seed2Tables Tg = ST2;
						const uint32_t or0 = ctx->s2_group_or[g], g_or = ctx->s2_group_nor[g];
						Tg.seeds = ST2.seeds + g_begin; Tg.n_seeds = ctx->s2_group_end[g] - g_begin;
						Tg.masks = ST2.masks + (size_t)or0; Tg.floors = ST2.floors + or0; Tg.n_or = g_or; Tg.or_base = or0;
						g_begin = ctx->s2_group_end[g];
						if(Tg.n_seeds == 0){ g_prefix += 1u; continue; }
						const size_t dyn = (size_t)g_or*sizeof(uint4) + (((size_t)g_or + 15) & ~size_t(15)) + 8*((size_t)Tg.n_seeds + 64) + 16;   // masks | floors | chain | head (each with 64 dummy slots)
						IrrArgs2 IA; IA.scan = S.irr_scan.p; IA.irr = S.irr.p; IA.n_live = n_live;
						IA.off_mask = ctx->s2_group_offmask[g];
						IA.exhaustive = first_launch ? 1u : 0u;                             // words holding IUPAC slots meet every candidate once, in the first launch
						IA.ix_first = IA.ix_last = IA.ix_words = nullptr; IA.min_cws = std::min<uint32_t>(min_oligo_length, 255u);
						if(!or_plain.empty()){                                              // unseeded candidates in the pass: every irregular word meets every candidate, once
							IA.off_mask = 0;
							if(!first_launch) IA.n_live = 0;
						}
						else if(irr_by_index){                                               // every candidate seeded, no IUPAC word in the set: the words come in through the index, by seed
							IA.ix_first = S.irx_first.p; IA.ix_last = S.irx_last.p; IA.ix_words = S.irx_words.p; IA.n_live = 0;
						}
						if(ctx->debug_log) fprintf(stderr, "[pcramp] k_seed2: %u workgroups, %u seeds of orientations %u..%u (group %zu of %zu), %zu + %zu B of LDS\n", sgrid.x, Tg.n_seeds,
							or0, or0 + g_or - 1, g + 1, ctx->s2_group_end.size(), sizeof(S2Shared), dyn);
						S2Clear Z = { nullptr, 0u, nullptr, 0u, nullptr };
						if(lean && !cleared_bits){                                           // the first launch of a lean pass clears the result bitsets
							Z.z0 = (uint4 *)fa->d_fr; Z.z1 = (uint4 *)fa->d_rf; Z.n0 = Z.n1 = (uint32_t)(lean_bits_bytes/16); Z.ctrl = d_counters;
							cleared_bits = true;
						}
						if(use_seed3){
							Seed3Tables T3; T3.seeds = Tg.seeds; T3.chunk_prefix = d_s3_prefix + g_prefix; T3.masks = Tg.masks; T3.floors = Tg.floors;
							g_prefix += Tg.n_seeds + 1u;
							T3.n_seeds = Tg.n_seeds; T3.n_or = Tg.n_or; T3.or_base = Tg.or_base;
							T3.pix_first = S.pix_first.p; T3.pix_last = S.pix_last.p; T3.pix_ent = S.pix_ent.p;
							Seed3Set Q3 = { S.valid_d(), S.blk_info.p, S.blk_local.p, S.d_active.p };
							int s3_wg = 0;
							const uint32_t G = seed3_grid(ctx, &s3_wg);
							const pcr_ctx::S3Launch &L3 = ctx->s3_launch[g];                   // (plan_seed3_slices)
							T3.n_chunks = L3.n_chunks; T3.per_wg = L3.per_wg; T3.slice_cap = L3.slice_cap;
							T3.n_irr_wg = IA.ix_first ? L3.n_irr_wg : 0u;           // (without the index the chunk workgroups are fewer than they could be: harmless)
							const Seed3Slices &W3 = L3.W;
							const size_t dyn3 = (size_t)g_or*sizeof(uint4) + 2*(size_t)T3.slice_cap*sizeof(uint32_t) + (((size_t)g_or + 15) & ~size_t(15)) + 16;
							if(!ctx->s3_attr_set){
								HIP_TRY(hipFuncSetAttribute((const void *)k_seed3<512>, hipFuncAttributeMaxDynamicSharedMemorySize, 128*1024));
								HIP_TRY(hipFuncSetAttribute((const void *)k_seed3<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 128*1024));
								ctx->s3_attr_set = true;
							}
							if(s3_wg == 1024) hipLaunchKernelGGL(k_seed3<1024>, dim3(G), dim3(1024), dyn3, ctx->stream, T3, Q3, IA, ctx->d_cand_fwd, ctx->d_cand_floor, sink, Z, W3);
							else hipLaunchKernelGGL(k_seed3<512>, dim3(G), dim3(512), dyn3, ctx->stream, T3, Q3, IA, ctx->d_cand_fwd, ctx->d_cand_floor, sink, Z, W3);
						}
						else if(ctx->s2_dbg)
							hipLaunchKernelGGL(k_seed2<true>, sgrid, sblock, dyn, ctx->stream, S.tb_d(), S.valid_d(), S.tile_desc.p, S.n_tiles, Tg, S.d_active.p, ctx->d_cand_fwd, ctx->d_cand_floor, ncand, IA, sink,
							ctx->s2_dbg, Z);
						else
							hipLaunchKernelGGL(k_seed2<false>, sgrid, sblock, dyn, ctx->stream, S.tb_d(), S.valid_d(), S.tile_desc.p, S.n_tiles, Tg, S.d_active.p, ctx->d_cand_fwd, ctx->d_cand_floor, ncand, IA, sink,
							ctx->s2_dbg, Z);
						HIP_TRY(hipGetLastError());
						first_launch = false;
					}
					if(need_seedset && (rc = launch_scan2(ctx, S, tab_seedset, ncand, sink, d_tab_seedset, d_bias_seedset, S.degen_tiles.p,
						S.n_degen_tiles, d_map_seedset)) != PCR_OK) return rc;
				}
				else if(!or_seed.empty()){
					const bool cand_lds = ncand <= SEED_CAND_LDS;   // (candidates read from global to fit 8 workgroups per CU: 140 vs 102 us)
					const size_t dyn = cand_lds ? (size_t)ncand*(2*sizeof(uint4) + sizeof(uint32_t)) : 0;
					// persistent workgroups: exactly as many as are resident at once (a partial second round would
					// run alone at the end), asked of the runtime for this kernel and its dynamic LDS size
					int per_cu = 0;
					const hipError_t oe = cand_lds ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_seed<true>, SEED_THREADS, dyn)
						: hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_seed<false>, SEED_THREADS, dyn);
					// The runtime's answer was one too high twice (7 for 21.6 KB of LDS, 6 for 27.2 KB: the extra workgroup
					// ran as a second round, 146 vs 109 us at C2); both cases fit "160 KB, allocated in 4 KB units".
					{
						const size_t lds_wg = ((sizeof(SeedShared) + dyn + 4095)/4096)*4096;
						per_cu = std::min<int>(per_cu, (int)((160*1024)/lds_wg));
						if(per_cu < 1) per_cu = 1;
					}
					const uint32_t resident = (oe == hipSuccess && per_cu > 0) ? (uint32_t)per_cu*ctx->n_cu : SEED_MAX_GRID;
					const uint32_t tiles_per_wg = SEED_WAVES*SEED_TILES_PER_WAVE;
					const dim3 sgrid(std::min<uint32_t>((S.n_tiles + tiles_per_wg - 1)/tiles_per_wg, resident)), sblock(SEED_THREADS);
					// the irregular words ride along as extra workgroups behind the persistent ones: they fill the
					// issue slots the latency-bound seed scan leaves idle instead of running alone afterwards
					IrrArgs IA; IA.irr = S.irr.p; IA.perm = S.irr_perm.p; IA.n_live = n_live; IA.off_mask = or_plain.empty() ? irr_off_mask : 0u;   // the seeded irregular scan needs every candidate seeded
					const uint32_t irr_wgs = (n_live + IRR_THREADS*IRR_PER_LANE - 1)/(IRR_THREADS*IRR_PER_LANE);
					const dim3 fgrid(sgrid.x + irr_wgs);
					irr_fused = true;
					if(ctx->debug_log) fprintf(stderr, "[pcramp] k_seed: %d workgroups per CU x %u CUs\n", per_cu, ctx->n_cu);
#define SEED_ARGS S.tb_d(), S.planes.p, S.valid_d(), S.d_blk_off.p, S.d_nblk_real.p, S.d_len.p, S.d_active.p, S.tile_seq.p, S.tile_pos0.p, \
	S.tile_degen.p, S.n_tiles, ST, ctx->d_cand_fwd, ctx->d_cand_rc, ctx->d_cand_floor, ncand, IA, sgrid.x, sink
					if(cand_lds) hipLaunchKernelGGL(k_seed<true>, fgrid, sblock, dyn, ctx->stream, SEED_ARGS);
					else hipLaunchKernelGGL(k_seed<false>, fgrid, sblock, dyn, ctx->stream, SEED_ARGS);
#undef SEED_ARGS
					HIP_TRY(hipGetLastError());
					if(need_seedset && (rc = launch_scan2(ctx, S, tab_seedset, ncand, sink, d_tab_seedset, d_bias_seedset, S.degen_tiles.p,
						S.n_degen_tiles, d_map_seedset)) != PCR_OK) return rc;
				}
			}
			scan_prof.finish();
		}
		if(n_live && !irr_fused){
			const unsigned irr_grid = (n_live + IRR_THREADS*IRR_PER_LANE - 1)/(IRR_THREADS*IRR_PER_LANE);
			hipLaunchKernelGGL(k_scan_irr, dim3(irr_grid), dim3(IRR_THREADS), 0, ctx->stream, S.irr.p, S.irr_perm.p, n_live,
				S.d_active.p, ctx->d_cand_fwd, ctx->d_cand_floor, ncand, sink);
			HIP_TRY(hipGetLastError());
		}
		if(async && fa && fa->staged && (cap == POST_CAP || cap == 128 || cap == 256) && 2*fa->n_pairs <= 32*POST_MASK_WORDS){
			// the whole tail -- DB finalisation and the amplicon screen -- in one launch
			++ctx->mail_seq;
			const uint64_t bw = (S.n + 63)/64;
#define POST_ARGS ctx->hits.p, d_seq_count, ctx->best.p, ncand, S.planes.p, S.d_blk_off.p, S.irr.p, S.irr_off.p, S.db.p, S.d_seg_hi, d_counters, ctx->epoch, S.n, \
	fa->d_oligos, fa->n_pairs, (2*fa->n_pairs + 31)/32, S.d_len.p, S.d_active.p, fa->a->amp_min, fa->a->amp_max, \
	fa->a->ident_threshold, fa->a->use_taq_mama, fa->d_fr, fa->d_rf, bw, ctx->mail_dev + (ctx->mail_seq % pcr_ctx::MAIL_RING), ctx->mail_seq
			if(cap == POST_CAP){
				static const int post_waves = getenv("PCRAMP_POST_WAVES") ? atoi(getenv("PCRAMP_POST_WAVES")) : 8;
				if(post_waves == 16) hipLaunchKernelGGL(k_post<16>, dim3((S.n + 15)/16), dim3(1024), 0, ctx->stream, POST_ARGS);
				else if(post_waves == 4) hipLaunchKernelGGL(k_post<4>, dim3((S.n + 3)/4), dim3(256), 0, ctx->stream, POST_ARGS);
				else hipLaunchKernelGGL(k_post<8>, dim3((S.n + 7)/8), dim3(512), 0, ctx->stream, POST_ARGS);
				S.ctrl_clean = true; S.touched_from_seg = true;              // k_post zeroes the counters and fills it has read
			}
			else if(cap == 128) hipLaunchKernelGGL((k_post_big<128, 4>), dim3((S.n + 3)/4), dim3(256), 0, ctx->stream, POST_ARGS);
			else hipLaunchKernelGGL((k_post_big<256, 4>), dim3((S.n + 3)/4), dim3(256), 0, ctx->stream, POST_ARGS);
#undef POST_ARGS
			HIP_TRY(hipGetLastError());
			fa->posted = true;
			S.db_cap = cap; S.n_slots = n_slots;
			S.n_touched = N_TOUCHED_UNKNOWN; S.n_entries = 1; S.have_db = true; S.touched_built = false;
			return PCR_OK;
		}
		hipLaunchKernelGGL(k_touched, dim3((S.n + 255)/256), dim3(256), 0, ctx->stream, d_seq_count, S.n, d_counters, S.touched.p);
		HIP_TRY(hipGetLastError());
		S.touched_built = true;
		uint32_t np2 = 1; while(np2 < cap) np2 <<= 1;
#define FIN_ARGS ctx->hits.p, d_seq_count, cap, ctx->best.p, ncand, S.planes.p, S.d_blk_off.p, S.irr.p, S.irr_off.p, S.db.p, S.d_seg_hi, \
	d_counters, ctx->epoch, S.touched.p
		if(np2 <= 1024) hipLaunchKernelGGL(k_finalize<FIN_WAVES>, dim3((S.n + FIN_WAVES - 1)/FIN_WAVES), dim3(64*FIN_WAVES), (size_t)FIN_WAVES*np2*sizeof(uint64_t), ctx->stream, FIN_ARGS);
		else if(np2 <= MAX_BUCKET_CAP) hipLaunchKernelGGL(k_finalize<1>, dim3(S.n), dim3(64), (size_t)np2*sizeof(uint64_t), ctx->stream, FIN_ARGS);
		else{
			if((rc = ctx->fin_scratch.ensure((size_t)S.n*np2)) != PCR_OK) return rc;
			hipLaunchKernelGGL(k_finalize_big, dim3(S.n), dim3(FINBIG_THREADS), 0, ctx->stream, FIN_ARGS, ctx->fin_scratch.p);
		}
#undef FIN_ARGS
		HIP_TRY(hipGetLastError());
		// the only host synchronisation of the pass: overflow flag + DB size, through the mapped mailbox
		++ctx->mail_seq;
		if(async && fa && fa->staged){ fa->pub_seq = ctx->mail_seq; fa->pub_counters = d_counters; }   // k_match publishes
		else{
			hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, ctx->stream, d_counters, ctx->mail_dev + (ctx->mail_seq % pcr_ctx::MAIL_RING), ctx->mail_seq);
			HIP_TRY(hipGetLastError());
		}
		if(async){
			S.db_cap = cap; S.n_slots = n_slots;
			S.n_touched = N_TOUCHED_UNKNOWN; S.n_entries = 1; S.have_db = true;
			return PCR_OK;
		}
		timer.next(3);
		if((rc = mail_wait(ctx, ctx->mail_seq, h_counters)) != PCR_OK) return rc;
		timer.next(2);
		S.db_cap = cap; S.n_slots = n_slots;
		if(!(h_counters[0] & 1u)){
			// Buckets grown for an earlier, denser pass over this set (a DB selected with every slot shift, a lower threshold) make
			// every consumer of this DB walk mostly empty slots (the local search: 3x slower on 2 048-slot buckets holding <= 200
			// entries; their slot-per-thread kernels start a thread per slot): when a quarter of them would do the pass is repeated
			// once with those.
			if(cap > 64 && attempt < 12){
				uint32_t want = 64;
				while(want < h_counters[2] + h_counters[2]/16) want *= 2;
				if(want*4 <= cap){ S.bucket_cap = want; continue; }
			}
			break;
		}
		// some sequence collected more hits than its bucket holds: grow the buckets and redo the pass
		if(attempt >= 12 || cap >= MAX_BUCKET_CAP_GLOBAL){ g_err = "pcr_select_words: more than 65536 candidate sites in one sequence (per-sequence bucket limit)"; return PCR_ERR_CAPACITY; }
		uint32_t want = cap*2;
		while(want < h_counters[2] && want < MAX_BUCKET_CAP_GLOBAL) want *= 2;
		S.bucket_cap = want;
	}
	if(ctx->debug_log) fprintf(stderr, "[pcramp] pass done: %u-slot buckets, largest fill %u, %u sequences with entries\n", S.db_cap, h_counters[2], h_counters[3]);
	S.n_touched = h_counters[3];
	S.n_entries = S.n_touched ? 1 : 0;   // "non-empty" marker; the exact count is taken on demand (count_entries)
	S.have_db = true;
	if(n_entries_out){
		if((rc = count_entries(ctx, S)) != PCR_OK) return rc;
		*n_entries_out = S.n_entries;
	}
	return PCR_OK;
}

// Look at the counters of the passes pcr_screen_device enqueued.  A pass whose buckets overflowed produced
// an incomplete DB (and so possibly incomplete amplification bits): grow the buckets and replay it and
// everything enqueued after it, synchronously, into the same output buffers.
int drain(pcr_ctx *ctx)
{
	if(ctx->pending.empty()) return PCR_OK;
	std::vector<pcr_ctx::Pending> pend;
	pend.swap(ctx->pending);
	int rc;
	for(size_t i = 0;i < pend.size();++i){
		uint32_t c[4];
		if((rc = mail_wait(ctx, pend[i].seq, c)) != PCR_OK) return rc;
		SeqSet &S = ctx->sets[pend[i].which];
		if(!(c[0] & 1u)){
			if(c[3] == N_TOUCHED_UNKNOWN_DEV){ S.n_touched = N_TOUCHED_UNKNOWN; S.n_entries = 1; }
			else{ S.n_touched = c[3]; S.n_entries = c[3] ? 1 : 0; }
			continue;
		}
		uint32_t want = S.bucket_cap*2;
		while(want < c[2] && want < MAX_BUCKET_CAP_GLOBAL) want *= 2;
		S.bucket_cap = std::min(want, MAX_BUCKET_CAP_GLOBAL);
		for(size_t j = i;j < pend.size();++j){
			const pcr_ctx::Pending &q = pend[j];
			if((rc = select_impl(ctx, (pcr_set)q.which, q.pairs.data(), (uint32_t)q.pairs.size(), q.opt5, q.opt3, q.thr, q.min_len, nullptr, false)) != PCR_OK) return rc;
			if((rc = amplify_launch(ctx, ctx->sets[q.which], q.pairs.data(), (uint32_t)q.pairs.size(), &q.args, q.d_fr, q.d_rf)) != PCR_OK) return rc;
		}
		break;
	}
	return PCR_OK;
}

} // namespace

extern "C" {

int pcr_select_words(pcr_ctx *ctx, pcr_set which, const pcr_pair *pairs, uint32_t n_pairs, int optimize_5, int optimize_3,
	float threshold, uint32_t min_oligo_length, uint64_t *n_entries_out)
{
	if(!set_ok(which)){ g_err = "pcr_select_words: unknown sequence set"; return PCR_ERR_ARG; }
	if(!ctx || (n_pairs && !pairs)){ g_err = "pcr_select_words: bad argument"; return PCR_ERR_ARG; }
	DRAIN(ctx);
	return select_impl(ctx, which, pairs, n_pairs, optimize_5, optimize_3, threshold, min_oligo_length, n_entries_out, false);
}

int pcr_screen_device(pcr_ctx *ctx, pcr_set which, const pcr_pair *pairs, uint32_t n_pairs, int optimize_5, int optimize_3,
	float select_threshold, uint32_t min_oligo_length, const pcr_amplify_args *args, uint64_t *d_bits_fr, uint64_t *d_bits_rf)
{
	if(!set_ok(which)){ g_err = "pcr_screen_device: unknown sequence set"; return PCR_ERR_ARG; }
	if(!ctx || !args || (n_pairs && (!pairs || !d_bits_fr || !d_bits_rf))){ g_err = "pcr_screen_device: bad argument"; return PCR_ERR_ARG; }
	if(ctx->pending.size() + 2 >= pcr_ctx::MAIL_RING) DRAIN(ctx);     // the mailbox ring bounds how far the host may run ahead
	const uint32_t seq0 = ctx->mail_seq;
	FusedAmp fa; fa.pairs = pairs; fa.n_pairs = n_pairs; fa.a = args; fa.d_fr = d_bits_fr; fa.d_rf = d_bits_rf;
	int rc = select_impl(ctx, which, pairs, n_pairs, optimize_5, optimize_3, select_threshold, min_oligo_length, nullptr, true, &fa);
	if(rc != PCR_OK) return rc;
	if(!fa.posted && (rc = amplify_launch(ctx, ctx->sets[which], pairs, n_pairs, args, d_bits_fr, d_bits_rf, &fa)) != PCR_OK) return rc;
	if(ctx->mail_seq != seq0){                                        // a pass was enqueued (not the empty-input shortcut)
		pcr_ctx::Pending p;
		p.seq = ctx->mail_seq; p.which = (int)which; p.pairs.assign(pairs, pairs + n_pairs); p.opt5 = optimize_5; p.opt3 = optimize_3;
		p.thr = select_threshold; p.min_len = min_oligo_length; p.args = *args; p.d_fr = d_bits_fr; p.d_rf = d_bits_rf;
		ctx->pending.push_back(p);
	}
	return PCR_OK;
}

int64_t pcr_get_entries(pcr_ctx *ctx, pcr_set which, pcr_entry *out, uint64_t cap)
{
	if(!set_ok(which)){ g_err = "pcr_get_entries: unknown sequence set"; return PCR_ERR_ARG; }
	if(!ctx){ g_err = "null ctx"; return PCR_ERR_ARG; }
	DRAIN(ctx);
	SeqSet &S = ctx->sets[which];
	if(!S.have_db){ g_err = "pcr_get_entries: no word DB"; return PCR_ERR_STATE; }
	{ const int rc = count_entries(ctx, S); if(rc != PCR_OK) return rc; }
	if(cap && out && S.n_entries){
		std::vector<DevEntry> h(S.n_slots);
		std::vector<uint32_t> hi(S.n);
		hipError_t e = hipStreamSynchronize(ctx->stream);
		if(e == hipSuccess) e = hipMemcpy(h.data(), S.db.p, S.n_slots*sizeof(DevEntry), hipMemcpyDeviceToHost);
		if(e == hipSuccess) e = hipMemcpy(hi.data(), S.d_seg_hi, S.n*sizeof(uint32_t), hipMemcpyDeviceToHost);
		if(e != hipSuccess){ g_err = std::string("pcr_get_entries: ") + hipGetErrorString(e); return PCR_ERR_DEVICE; }
		uint64_t k = 0;
		for(uint32_t q = 0;q < S.n && k < cap;++q){
			for(uint32_t i = q*S.db_cap;i < hi[q] && k < cap;++i, ++k){
				pcrhost::word_of_planes(h[i].w, out[k].word.w);
				out[k].loc = h[i].loc; out[k].index = h[i].seq; out[k].strand = h[i].strand; out[k].pad = 0;
			}
		}
	}
	return (int64_t)S.n_entries;
}

int pcr_amplify_device(pcr_ctx *ctx, pcr_set which, const pcr_pair *pairs, uint32_t n_pairs, const pcr_amplify_args *args,
	uint64_t *d_bits_fr, uint64_t *d_bits_rf)
{
	if(!set_ok(which)){ g_err = "pcr_amplify_device: unknown sequence set"; return PCR_ERR_ARG; }
	if(!ctx || !args || (n_pairs && (!pairs || !d_bits_fr || !d_bits_rf))){ g_err = "pcr_amplify_device: bad argument"; return PCR_ERR_ARG; }
	DRAIN(ctx);
	HIP_TRY(hipSetDevice(ctx->device));
	return amplify_launch(ctx, ctx->sets[which], pairs, n_pairs, args, d_bits_fr, d_bits_rf);
}

int pcr_amplify(pcr_ctx *ctx, pcr_set which, const pcr_pair *pairs, uint32_t n_pairs, const pcr_amplify_args *args,
	uint64_t *bits, uint64_t *bits_fr, uint64_t *bits_rf, float *coverage)
{
	if(!set_ok(which)){ g_err = "pcr_amplify: unknown sequence set"; return PCR_ERR_ARG; }
	if(!ctx || !args || (n_pairs && !pairs)){ g_err = "pcr_amplify: bad argument"; return PCR_ERR_ARG; }
	DRAIN(ctx);
	HIP_TRY(hipSetDevice(ctx->device));
	SeqSet &S = ctx->sets[which];
	const uint64_t words = (S.n + 63)/64;
	const size_t total = (size_t)n_pairs*words;
	int rc;
	if((rc = ctx->bits_fr.ensure(total + 2)) != PCR_OK) return rc;               // + 2: return_to_host copies whole 16-byte units
	if((rc = ctx->bits_rf.ensure(total + 2)) != PCR_OK) return rc;
	if((rc = ctx->status.ensure(1)) != PCR_OK) return rc;
	if((rc = amplify_launch(ctx, S, pairs, n_pairs, args, ctx->bits_fr.p, ctx->bits_rf.p)) != PCR_OK) return rc;
	const bool have_status = S.n_entries && n_pairs;                             // the screen ran and wrote the status word
	const uint8_t *back[3];
	if((rc = return_to_host(ctx, ctx->bits_fr.p, total*sizeof(uint64_t), ctx->bits_rf.p, total*sizeof(uint64_t),
		have_status ? ctx->status.p : nullptr, have_status ? sizeof(uint32_t) : 0, back)) != PCR_OK) return rc;
	const uint64_t *hfr = (const uint64_t *)back[0], *hrf = (const uint64_t *)back[1];
	const uint32_t status = have_status ? *(const uint32_t *)back[2] : 0u;
	if(status & 1u){ g_err = "Sequence::has_split: range is out of bounds"; return PCR_ERR_RANGE; }   // sequence.cpp:306-308
	for(uint32_t p = 0;p < n_pairs;++p){
		if(bits){ for(uint64_t w = 0;w < words;++w) bits[p*words + w] = hfr[p*words + w] | hrf[p*words + w]; }
		if(bits_fr) memcpy(bits_fr + p*words, hfr + p*words, words*sizeof(uint64_t));
		if(bits_rf) memcpy(bits_rf + p*words, hrf + p*words, words*sizeof(uint64_t));
		if(coverage) coverage[p] = pcr_coverage_from_bits(hfr + p*words, hrf + p*words, S.weight.data(), S.n);
	}
	return PCR_OK;
}

int pcr_move_coverage(pcr_ctx *ctx, pcr_set which, const pcr_pair *base, int side, const pcr_word128 *variants, uint32_t n_variants,
	const pcr_amplify_args *args, uint64_t *bits_fr, uint64_t *bits_rf, float *coverage)
{
	if(!set_ok(which)){ g_err = "pcr_move_coverage: unknown sequence set"; return PCR_ERR_ARG; }
	if(!ctx || !args || !base || (side != 0 && side != 1) || (n_variants && !variants)){ g_err = "pcr_move_coverage: bad argument"; return PCR_ERR_ARG; }
	DRAIN(ctx);
	HIP_TRY(hipSetDevice(ctx->device));
	SeqSet &S = ctx->sets[which];
	if(!S.have_db){ g_err = "pcr_move_coverage: no word DB (call pcr_select_words first)"; return PCR_ERR_STATE; }
	{ const int erc = ensure_touched(ctx, S); if(erc != PCR_OK) return erc; }
	const uint64_t words = (S.n + 63)/64;
	const size_t total = (size_t)n_variants*words;
	if(coverage){ for(uint32_t v = 0;v < n_variants;++v) coverage[v] = 0.0f; }
	if(bits_fr && total) memset(bits_fr, 0, total*sizeof(uint64_t));
	if(bits_rf && total) memset(bits_rf, 0, total*sizeof(uint64_t));
	if(S.n_entries == 0 || n_variants == 0) return PCR_OK;
	int rc;
	// oligo table: base F, base R (floors from the collect threshold, pcr_assay.cpp:31-32), then the variants
	const float thr2 = args->collect_threshold*args->collect_threshold;
	std::vector<OligoDev> ol(2 + (size_t)n_variants);
	fill_oligo(ol[0], base->f.w, thr2);
	fill_oligo(ol[1], base->r.w, thr2);
	for(uint32_t v = 0;v < n_variants;++v){
		if((variants[v].w[0] | variants[v].w[1]) == 0){ g_err = "pcr_move_coverage: empty trial oligo"; return PCR_ERR_ARG; }
		fill_oligo(ol[2 + v], variants[v].w, thr2);
	}
	if((rc = ctx->bits_fr.ensure(total + 2)) != PCR_OK) return rc;
	if((rc = ctx->bits_rf.ensure(total + 2)) != PCR_OK) return rc;
	Stager st(ctx);
	if((rc = st.begin(ol.size()*sizeof(OligoDev) + 64)) != PCR_OK) return rc;
	const OligoDev *d_ol = st.put(ol.data(), ol.size());
	if((rc = st.ship(ctx->bits_fr.p, total*sizeof(uint64_t), ctx->bits_rf.p, total*sizeof(uint64_t))) != PCR_OK) return rc;
	if((rc = ctx->mask.ensure((size_t)S.n_slots)) != PCR_OK) return rc;
	if((rc = ctx->status.ensure(1)) != PCR_OK) return rc;
	const uint32_t n_db = S.n_touched*S.db_cap;
	hipLaunchKernelGGL(k_match_few, dim3((n_db + 255)/256), dim3(256), 0, ctx->stream, S.db.p, n_db, S.db_cap, S.touched.p, S.d_seg_hi, d_ol, 2u,
		ctx->mask.p, ctx->status.p);
	HIP_TRY(hipGetLastError());
	if(S.db_cap >= 256 && S.db_cap <= PMS_MAX)                                    // big buckets: a workgroup per sequence over a compact partner list
		hipLaunchKernelGGL(k_pair_moves_seq, dim3(S.n_touched), dim3(PMS_THREADS), 0, ctx->stream, S.db.p, S.db_cap, S.touched.p, S.d_seg_hi, ctx->mask.p,
			d_ol, d_ol + 2, n_variants, side, S.planes.p, S.d_blk_off.p, S.d_len.p, S.d_active.p, S.d_has_eos.p, args->amp_min, args->amp_max,
			args->ident_threshold, args->use_taq_mama, ctx->bits_fr.p, ctx->bits_rf.p, words, ctx->status.p);
	else
		hipLaunchKernelGGL(k_pair_moves, dim3((n_db + 127)/128), dim3(128), 0, ctx->stream, S.db.p, n_db, S.db_cap, S.touched.p, S.d_seg_hi, ctx->mask.p,
			d_ol, d_ol + 2, n_variants, side, S.planes.p, S.d_blk_off.p, S.d_len.p, S.d_active.p, S.d_has_eos.p, args->amp_min, args->amp_max,
			args->ident_threshold, args->use_taq_mama, ctx->bits_fr.p, ctx->bits_rf.p, words, ctx->status.p);
	HIP_TRY(hipGetLastError());
	const uint8_t *back[3];
	if((rc = return_to_host(ctx, ctx->bits_fr.p, total*sizeof(uint64_t), ctx->bits_rf.p, total*sizeof(uint64_t), ctx->status.p, sizeof(uint32_t), back)) != PCR_OK) return rc;
	const uint64_t *hfr = (const uint64_t *)back[0], *hrf = (const uint64_t *)back[1];
	const uint32_t status = *(const uint32_t *)back[2];
	if(status & 1u){ g_err = "Sequence::has_split: range is out of bounds"; return PCR_ERR_RANGE; }   // sequence.cpp:306-308
	if(bits_fr) memcpy(bits_fr, hfr, total*sizeof(uint64_t));
	if(bits_rf) memcpy(bits_rf, hrf, total*sizeof(uint64_t));
	if(coverage){ for(uint32_t v = 0;v < n_variants;++v) coverage[v] = pcr_coverage_from_bits(hfr + v*words, hrf + v*words, S.weight.data(), S.n); }
	return PCR_OK;
}

float pcr_coverage_from_bits(const uint64_t *bits_fr, const uint64_t *bits_rf, const float *weights, uint64_t n)
{
	// PCR::compute_coverage (pcr_assay.cpp:271-302): amplicons are visited {F(+),R(-)} first in
	// ascending sequence order, then {R(+),F(-)}; each sequence's weight is added once, in double.
	double ret = 0.0;
	const uint64_t nw = (n + 63)/64;
	for(uint64_t w = 0;w < nw;++w){                                               // set bits in ascending order
		for(uint64_t m = bits_fr[w];m;m &= m - 1){ const uint64_t i = w*64 + (uint64_t)__builtin_ctzll(m); if(i < n) ret += weights[i]; }
	}
	for(uint64_t w = 0;w < nw;++w){
		for(uint64_t m = bits_rf[w] & ~bits_fr[w];m;m &= m - 1){ const uint64_t i = w*64 + (uint64_t)__builtin_ctzll(m); if(i < n) ret += weights[i]; }
	}
	return (float)ret;
}

float pcr_weighted_coverage(const uint64_t *bits, const float *weights, uint64_t n)
{
	double ret = 0.0;                                                            // main.cpp:1402-1418
	const uint64_t nw = (n + 63)/64;
	for(uint64_t w = 0;w < nw;++w){
		for(uint64_t m = bits[w];m;m &= m - 1){ const uint64_t i = w*64 + (uint64_t)__builtin_ctzll(m); if(i < n) ret += weights[i]; }
	}
	return (float)ret;
}

int pcr_profile_enable(pcr_ctx *ctx, int on)
{
	if(!ctx){ g_err = "null ctx"; return PCR_ERR_ARG; }
	ctx->prof = (on != 0);
	ctx->prof_stride = (on > 1) ? (uint32_t)on : 1u; ctx->prof_pass = 0;
	return PCR_OK;
}

int pcr_profile_read(pcr_ctx *ctx, double *scan_ms, uint64_t *scan_launches, int reset)
{
	if(!ctx){ g_err = "null ctx"; return PCR_ERR_ARG; }
	HIP_TRY(hipStreamSynchronize(ctx->stream));
	for(auto &pr : ctx->prof_events){
		float ms = 0.0f;
		HIP_TRY(hipEventElapsedTime(&ms, pr.first, pr.second));
		ctx->prof_ms += ms; ctx->prof_launches += 1;
		(void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second);
	}
	ctx->prof_events.clear();
	if(scan_ms) *scan_ms = ctx->prof_ms;
	if(scan_launches) *scan_launches = ctx->prof_launches;
	if(reset){ ctx->prof_ms = 0.0; ctx->prof_launches = 0; }
	return PCR_OK;
}

int pcr_profile_read_kernel(pcr_ctx *ctx, int kernel, double *ms, uint64_t *launches, int reset)
{
	if(!ctx || kernel < 0 || kernel >= PCR_PROF_KERNELS){ g_err = "pcr_profile_read_kernel: bad argument"; return PCR_ERR_ARG; }
	if(kernel == PCR_PROF_SCAN) return pcr_profile_read(ctx, ms, launches, reset);
	HIP_TRY(hipStreamSynchronize(ctx->stream));
	for(auto &pr : ctx->prof_events_k[kernel]){
		float t = 0.0f;
		HIP_TRY(hipEventElapsedTime(&t, pr.first, pr.second));
		ctx->prof_ms_k[kernel] += t; ctx->prof_launches_k[kernel] += 1;
		(void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second);
	}
	ctx->prof_events_k[kernel].clear();
	if(ms) *ms = ctx->prof_ms_k[kernel];
	if(launches) *launches = ctx->prof_launches_k[kernel];
	if(reset){ ctx->prof_ms_k[kernel] = 0.0; ctx->prof_launches_k[kernel] = 0; }
	return PCR_OK;
}

// ------------------------------------------------------------------ host-only helpers
static pcrhost::PackFilter make_filter(const pcr_params *params)
{
	pcrhost::PackFilter f;
	f.max_degen = params ? params->pack_max_degen : 256;
	f.set_gc(params ? params->pack_min_gc : 0.0f, params ? params->pack_max_gc : 1.0f);
	return f;
}

int64_t pcr_host_irregular_words(const uint8_t *packed4, uint64_t len, const pcr_params *params,
	uint32_t min_oligo_length, pcr_entry *out, uint64_t cap)
{
	if(len && !packed4){ g_err = "pcr_host_irregular_words: bad argument"; return PCR_ERR_ARG; }
	std::vector<uint8_t> buf(packed4, packed4 + (len + 1)/2);
	if((len & 1) && !buf.empty()) buf.back() &= 0xF0;
	const pcrhost::PackFilter f = make_filter(params);
	std::vector<pcrhost::IrrEntry> v;
	pcrhost::PackedSeq q; q.buf = buf.data(); q.len = len;
	if(!pcrhost::irregular_words(q, f, v)){ g_err = "irregular word overflow"; return PCR_ERR_CAPACITY; }
	uint64_t n = 0;
	for(const pcrhost::IrrEntry &e : v){
		if(e.cws < min_oligo_length) continue;
		if(n < cap && out){
			pcrhost::word_of_planes(e.w, out[n].word.w);
			out[n].loc = e.loc; out[n].index = 0; out[n].strand = e.strand; out[n].pad = e.ord;
		}
		++n;
	}
	return (int64_t)n;
}

int pcr_host_window_valid(const uint8_t *packed4, uint64_t len, const pcr_params *params, uint8_t *valid_out)
{
	if(len && (!packed4 || !valid_out)){ g_err = "pcr_host_window_valid: bad argument"; return PCR_ERR_ARG; }
	const pcrhost::PackFilter f = make_filter(params);
	pcrhost::PackedSeq q; q.buf = packed4; q.len = len;
	for(uint64_t p = 0;p < len;++p){
		bool ok = (p + 32 <= len);
		unsigned n2 = 0, n3 = 0, n4 = 0, ngc = 0;
		for(int k = 0;ok && k < 32;++k){
			const unsigned v = q.at(p + k);
			if(v == 0){ ok = false; break; }
			const int d = __builtin_popcount(v);
			n2 += (d == 2); n3 += (d == 3); n4 += (d == 4);
			ngc += ((v & 6) != 0);
		}
		if(ok && !((f.gc_ok >> ngc) & 1)) ok = false;
		if(ok && pcrhost::degeneracy_exceeds(n2, n3, n4, f.max_degen)) ok = false;
		valid_out[p] = ok ? 1 : 0;
	}
	return PCR_OK;
}

int64_t pcr_host_candidates(const pcr_pair *pairs, uint32_t n_pairs, int optimize_5, int optimize_3,
	float threshold, pcr_word128 *words_out, uint32_t *floors_out, uint64_t cap)
{
	std::vector<pcrhost::Candidate> cand;
	pcrhost::build_candidates((const uint64_t *)pairs, n_pairs, optimize_5 != 0, optimize_3 != 0, threshold, cand);
	for(size_t i = 0;i < cand.size() && i < cap;++i){
		if(words_out) pcrhost::word_of_planes(cand[i].fwd, words_out[i].w);
		if(floors_out) floors_out[i] = cand[i].floor_;
	}
	return (int64_t)cand.size();
}

int64_t pcr_host_orientation_seeds(const pcr_word128 *oligo, uint32_t floor, uint32_t *codes, uint8_t *q, uint8_t *off, uint64_t cap)
{
	if(!oligo){ g_err = "null oligo"; return PCR_ERR_ARG; }
	std::vector<pcrhost::Seed> seeds;
	if(!pcrhost::orientation_seeds(pcrhost::planes_of_word(oligo->w), floor, 0, seeds)) return -1;
	for(size_t i = 0;i < seeds.size() && i < cap;++i){
		if(codes) codes[i] = seeds[i].code;
		if(q) q[i] = seeds[i].q;
		if(off) off[i] = seeds[i].off;
	}
	return (int64_t)seeds.size();
}

int64_t pcr_host_move_trials(const pcr_word128 *oligo, int move, double max_degen, int primer_min, int primer_max,
	pcr_word128 *trials_out, uint64_t cap)
{
	if(!oligo){ g_err = "null oligo"; return PCR_ERR_ARG; }
	std::vector<Planes> t;
	if(!pcrhost::move_trials(pcrhost::planes_of_word(oligo->w), move, max_degen, primer_min, primer_max, t)){
		g_err = "pcr_host_move_trials: unknown move"; return PCR_ERR_ARG;
	}
	for(size_t i = 0;i < t.size() && i < cap;++i) pcrhost::word_of_planes(t[i], trials_out[i].w);
	return (int64_t)t.size();
}

} // extern "C"

#include "pcr_entry_sw_thermo.inc"
#include "pcr_multiplex_screen.inc"
#include "pcr_writers.inc"
#include "pcr_exchange.inc"
