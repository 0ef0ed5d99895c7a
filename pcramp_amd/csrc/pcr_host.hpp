// Host-side data model of the MI355X evaluation path (plain C++, no HIP).
//
// The device never materialises the reference's word index (Sequence::pack, sequence.cpp:92-267
// emits ~2L (Word, WordMatch) pairs per sequence per design iteration).  Instead a sequence is
// split into
//   * REGULAR windows: every start p whose 32 bases p..p+31 are non-EOS.  pack emits exactly one
//     plus and one minus full word for such a window (sequence.cpp:181-194) unless the window
//     degeneracy / GC filters skip it (sequence.cpp:127-153).  These are scanned on the GPU straight
//     from the bit-plane store; and
//   * IRREGULAR words: everything else pack emits -- the centred partial words at the head
//     (sequence.cpp:155-179) and tail (sequence.cpp:198-263), and whatever the streaming window
//     produces within 31 pushes of an EOS nibble.  There are O(32) of them per sequence end / EOS;
//     they are produced here by running the streaming state machine only over those zones and
//     are kept as an explicit word list.
// DESIGN.md "Exactness of the window model" has the argument for why the split is exact.
#ifndef PCR_HOST_HPP
#define PCR_HOST_HPP

#include <stdint.h>
#include <algorithm>
#include <deque>
#include <vector>

namespace pcrhost {

// A 32-slot oligo / word as four slot masks: bit k of a = slot k may be 'A', etc.
// (slot 0 = 5' end).  An EOS slot has no bit in any plane.
struct Planes {
	uint32_t a, c, g, t;
};

inline uint32_t bitrev32(uint32_t v)
{
	v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
	v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
	v = ((v >> 4) & 0x0F0F0F0Fu) | ((v & 0x0F0F0F0Fu) << 4);
	v = ((v >> 8) & 0x00FF00FFu) | ((v & 0x00FF00FFu) << 8);
	return (v >> 16) | (v << 16);
}

// Reference Word layout (word.cpp:11-16) <-> slot nibbles.
inline unsigned word_get(const uint64_t w[2], int k) { return (unsigned)(w[k >> 4] >> ((15 - (k & 15))*4)) & 0xF; }

inline void word_set(uint64_t w[2], int k, unsigned v)
{
	const int sh = (15 - (k & 15))*4;
	w[k >> 4] = (w[k >> 4] & ~(uint64_t(0xF) << sh)) | (uint64_t(v) << sh);
}

// Bits 0, 4, 8, ... 60 of x gathered into bits 0 ... 15.
inline uint32_t gather_every_4th(uint64_t x)
{
	x &= 0x1111111111111111ull;
	x = (x | (x >> 3)) & 0x0303030303030303ull;
	x = (x | (x >> 6)) & 0x000F000F000F000Full;
	x = (x | (x >> 12)) & 0x000000FF000000FFull;
	return (uint32_t)((x | (x >> 24)) & 0xFFFFu);
}

inline Planes planes_of_word(const uint64_t w[2])
{
	// slot k of a block is nibble 15 - (k & 15) (word_get): reverse the nibbles once, then plane j = every 4th bit from bit j
	uint64_t r[2];
	for(int h = 0;h < 2;++h){
		const uint64_t b = __builtin_bswap64(w[h]);
		r[h] = ((b >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((b & 0x0F0F0F0F0F0F0F0Full) << 4);
	}
	Planes p;
	p.a = gather_every_4th(r[0]) | (gather_every_4th(r[1]) << 16);
	p.c = gather_every_4th(r[0] >> 1) | (gather_every_4th(r[1] >> 1) << 16);
	p.g = gather_every_4th(r[0] >> 2) | (gather_every_4th(r[1] >> 2) << 16);
	p.t = gather_every_4th(r[0] >> 3) | (gather_every_4th(r[1] >> 3) << 16);
	return p;
}

inline void word_of_planes(const Planes &p, uint64_t w[2])
{
	w[0] = w[1] = 0;
	for(int k = 0;k < 32;++k){
		const unsigned v = ((p.a >> k) & 1) | (((p.c >> k) & 1) << 1) | (((p.g >> k) & 1) << 2) | (((p.t >> k) & 1) << 3);
		word_set(w, k, v);
	}
}

// The reverse complement in slot space: slot k -> slot 31-k, A<->T, C<->G.  Scanning the plus
// strand with these masks counts what the reference counts against the minus-strand word
// Word::complement() of the same window (word.h:140-183, sequence.cpp:188-190).
inline Planes planes_revcomp(const Planes &p)
{
	Planes r;
	r.a = bitrev32(p.t);
	r.t = bitrev32(p.a);
	r.c = bitrev32(p.g);
	r.g = bitrev32(p.c);
	return r;
}

inline uint32_t planes_occupied(const Planes &p) { return p.a | p.c | p.g | p.t; }
inline int planes_size(const Planes &p) { return __builtin_popcount(planes_occupied(p)); }                 // Word::size, word.cpp:199
inline int planes_start(const Planes &p) { const uint32_t o = planes_occupied(p); return o ? __builtin_ctz(o) : 32; } // word.h:256
inline int planes_stop(const Planes &p) { const uint32_t o = planes_occupied(p); return o ? 31 - __builtin_clz(o) : -1; } // word.h:273

inline unsigned planes_nibble(const Planes &p, int k)
{
	return ((p.a >> k) & 1) | (((p.c >> k) & 1) << 1) | (((p.g >> k) & 1) << 2) | (((p.t >> k) & 1) << 3);
}

// Word::shift_left / shift_right (word.cpp:215-231): towards slot 0 / towards slot 31.
inline Planes planes_shift_left(const Planes &p) { Planes r = {p.a >> 1, p.c >> 1, p.g >> 1, p.t >> 1}; return r; }
inline Planes planes_shift_right(const Planes &p) { Planes r = {p.a << 1, p.c << 1, p.g << 1, p.t << 1}; return r; }

// ---------------------------------------------------------------------------------------------
// Window filters of Sequence::pack, evaluated for a full 32-base window.
struct PackFilter {
	uint32_t max_degen;   // m_degen_pack_threshold
	uint64_t gc_ok;       // bit n set <=> a window with n G/C-containing slots passes (all ones if filter off)
	bool gc_on;

	// sequence.cpp:102-106,127-146: fraction = num_gc * (1.0f/32) compared in float.
	void set_gc(float min_gc, float max_gc)
	{
		gc_on = (min_gc > 0.0f) || (max_gc < 1.0f);
		gc_ok = 0;
		const float norm = 1.0f/32;
		for(unsigned n = 0;n <= 32;++n){
			const float f = n*norm;
			const bool skip = gc_on && ((f < min_gc) || (f > max_gc));
			if(!skip){ gc_ok |= (uint64_t(1) << n); }
		}
	}
};

// Word::degeneracy() > threshold (word.h:97-138, sequence.cpp:149) for a window with n2/n3/n4
// slots of 2/3/4-fold degeneracy.  The product 2^n2 * 3^n3 * 4^n4 is exact in double (3^32 < 2^53),
// so the comparison is an exact integer one: 3^n3 * 2^e > thr  <=>  3^n3 > floor(thr / 2^e).
inline bool degeneracy_exceeds(unsigned n2, unsigned n3, unsigned n4, uint32_t thr)
{
	const unsigned e = n2 + 2*n4;
	if(e >= 33){ return true; }
	uint64_t p3 = 1;
	for(unsigned i = 0;i < n3;++i){ p3 *= 3; }
	return p3 > (uint64_t(thr) >> e);
}

// ---------------------------------------------------------------------------------------------
// Irregular words.
struct IrrEntry {
	Planes w;        // the DB word exactly as pack stores it (minus-strand words already complemented)
	int32_t loc;     // WordMatch::loc
	uint8_t strand;  // 1 plus, 2 minus
	uint8_t cws;     // the size counter pack compares with m_min_oligo_length (32 = full-word emission)
	uint8_t ord;     // tie-break among irregular entries sharing (loc, strand) in one sequence
	uint8_t pad;
};

struct PackedSeq {
	const uint8_t *buf;   // high nibble first
	uint64_t len;         // bases
	unsigned at(uint64_t i) const
	{
		if(i >= len){ return 0; }   // the odd-length pad nibble is EOS (sequence.cpp:21)
		const uint8_t v = buf[i >> 1];
		return (i & 1) ? (v & 0xF) : (v >> 4);
	}
};

// The streaming window of Sequence::pack (sequence.cpp:92-267) as an explicit state machine
// over slot nibbles.
class PackMachine {
public:
	PackMachine(const PackFilter &f, std::vector<IrrEntry> &out) : filt(f), dst(out) { reset(); }

	void reset()
	{
		for(int k = 0;k < 32;++k){ s[k] = 0; }
		cws = 0;
		gc.clear();
		num_gc = 0;
	}

	// State right after a REGULAR iteration whose window is bases[0..31].
	void seed_after_regular(const uint8_t bases[32])
	{
		for(int k = 0;k < 32;++k){ s[k] = bases[k]; }
		cws = 31;
		gc.clear();
		num_gc = 0;
		for(int k = 0;k < 32;++k){ gc.push_back(bases[k]); num_gc += ((bases[k] & 6) != 0); }
	}

	// One loop iteration of sequence.cpp:110-195 with `loc` as the reference counts it (1-based).
	// `record` = false suppresses the output (used for regular iterations, which the GPU owns).
	void push(unsigned b, int loc, bool record)
	{
		push_back(b);
		cws += (b != 0);
		if(filt.gc_on){
			if(gc.size() == 32){ num_gc -= ((gc.front() & 6) != 0); gc.pop_front(); }
			gc.push_back((uint8_t)b);
			num_gc += ((b & 6) != 0);
			if(!((filt.gc_ok >> num_gc) & 1)){ cws = std::min(cws, 31u); return; }
		}
		if(degenerate()){ cws = std::min(cws, 31u); return; }
		if(cws < 32){
			if(record){ emit_partial(loc - int(cws), loc - 1); }
		}
		else{
			if(record){
				uint8_t rc[32];
				revcomp(s, rc);
				emit(s, loc - 32, 1, 32);
				emit(rc, loc - 1, 2, 32);
			}
			--cws;
		}
	}

	// The trailing loop, sequence.cpp:198-263; `loc` is the loop counter's final value.
	void drain(int loc)
	{
		while(cws > 0){
			for(int k = 0;k < 31;++k){ s[k] = s[k + 1]; }
			s[31] = 0;
			--cws;
			if(filt.gc_on){
				if(gc.size() == 32){ num_gc -= ((gc.front() & 6) != 0); gc.pop_front(); }
				if(!((filt.gc_ok >> num_gc) & 1)){ continue; }
			}
			if(degenerate()){ continue; }
			emit_partial(loc - 1 - int(cws), loc - 2);
		}
	}

private:
	const PackFilter &filt;
	std::vector<IrrEntry> &dst;
	uint8_t s[32];
	unsigned cws;
	std::deque<uint8_t> gc;
	unsigned num_gc;

	int start_of(const uint8_t *x) const { for(int k = 0;k < 32;++k){ if(x[k]) return k; } return 32; }
	int stop_of(const uint8_t *x) const { for(int k = 31;k >= 0;--k){ if(x[k]) return k; } return -1; }

	void push_back(unsigned b)                                                    // word.cpp:31-48
	{
		const int last = stop_of(s) + 1;
		if(last < 32){ s[last] = (uint8_t)b; return; }
		for(int k = 0;k < 31;++k){ s[k] = s[k + 1]; }
		s[31] = (uint8_t)b;
	}

	bool degenerate() const
	{
		unsigned n2 = 0, n3 = 0, n4 = 0;
		for(int k = 0;k < 32;++k){
			const int d = __builtin_popcount(s[k]);
			n2 += (d == 2); n3 += (d == 3); n4 += (d == 4);
		}
		return degeneracy_exceeds(n2, n3, n4, filt.max_degen);
	}

	static void center(uint8_t *x)                                                // word.h:392-418
	{
		int left = 32, right = -1;
		for(int k = 0;k < 32;++k){ if(x[k]){ left = k; break; } }
		for(int k = 31;k >= 0;--k){ if(x[k]){ right = k; break; } }
		if(left > right){ return; }
		right = 32 - right;
		const int delta = (right - left)/2;
		uint8_t y[32];
		for(int k = 0;k < 32;++k){
			const int src = k - delta;
			y[k] = (src >= 0 && src < 32) ? x[src] : 0;
		}
		for(int k = 0;k < 32;++k){ x[k] = y[k]; }
	}

	void revcomp(const uint8_t *x, uint8_t *y) const                              // word.h:140-183
	{
		for(int k = 0;k < 32;++k){ y[k] = 0; }
		const int first = start_of(x), last = stop_of(x);
		int d = 0;
		for(int src = last;src >= first;--src, ++d){
			const unsigned v = x[src];
			y[d] = (uint8_t)(((v & 1) << 3) | ((v & 8) >> 3) | ((v & 2) << 1) | ((v & 4) >> 1));
		}
	}

	// sequence.cpp:157-178 / 241-255: centre, store plus; complement, re-centre, store minus.
	void emit_partial(int plus_base, int minus_base)
	{
		uint8_t t[32], u[32];
		for(int k = 0;k < 32;++k){ t[k] = s[k]; }
		center(t);
		if(cws >= 1){ emit(t, plus_base - start_of(t), 1, (uint8_t)cws); }
		revcomp(t, u);
		center(u);
		if(cws >= 1){ emit(u, minus_base + start_of(u), 2, (uint8_t)cws); }
	}

	void emit(const uint8_t *x, int loc, uint8_t strand, uint8_t size_counter)
	{
		IrrEntry e;
		e.w.a = e.w.c = e.w.g = e.w.t = 0;
		for(int k = 0;k < 32;++k){
			e.w.a |= uint32_t(x[k] & 1) << k;
			e.w.c |= uint32_t((x[k] >> 1) & 1) << k;
			e.w.g |= uint32_t((x[k] >> 2) & 1) << k;
			e.w.t |= uint32_t((x[k] >> 3) & 1) << k;
		}
		e.loc = loc;
		e.strand = strand;
		e.cws = size_counter;
		e.ord = 0;
		e.pad = 0;
		dst.push_back(e);
	}
};

// All irregular words of one sequence, in emission order, with `ord` filled in.
// Returns false if a tie-break ordinal would not fit (never seen in practice: > 63 irregular
// words sharing one (loc, strand)).
inline bool irregular_words(const PackedSeq &q, const PackFilter &filt, std::vector<IrrEntry> &out)
{
	const size_t first_out = out.size();
	const uint64_t L = q.len;
	const uint64_t n_iter = (L + 1) & ~uint64_t(1);   // the loop walks whole bytes (sequence.cpp:110-120)
	if(n_iter == 0){ return true; }

	// Irregular iteration ranges [lo, hi] (0-based iteration = loc - 1): the head, and 32
	// iterations from every EOS nibble (the pad nibble of an odd-length sequence included).
	std::vector<std::pair<uint64_t, uint64_t> > zones;
	zones.push_back(std::make_pair(uint64_t(0), std::min<uint64_t>(30, n_iter - 1)));
	const uint64_t nbytes = n_iter/2;
	for(uint64_t byte = 0;byte < nbytes;++byte){
		const uint8_t v = q.buf[byte];
		const bool hi_zero = ((v & 0xF0) == 0);
		const bool lo_zero = ((v & 0x0F) == 0) || (2*byte + 1 >= L);
		if(!hi_zero && !lo_zero){ continue; }
		for(int h = 0;h < 2;++h){
			if(h == 0 ? !hi_zero : !lo_zero){ continue; }
			const uint64_t e = 2*byte + h;
			const uint64_t hi = std::min<uint64_t>(e + 31, n_iter - 1);
			if(!zones.empty() && e <= zones.back().second + 1){ zones.back().second = std::max(zones.back().second, hi); }
			else{ zones.push_back(std::make_pair(e, hi)); }
		}
	}

	PackMachine m(filt, out);
	bool have_state = false;    // machine state is valid for the iteration following `state_at`
	uint64_t state_next = 0;
	for(size_t z = 0;z < zones.size();++z){
		const uint64_t lo = zones[z].first, hi = zones[z].second;
		if(lo == 0){ m.reset(); }
		else if(!(have_state && state_next == lo)){
			uint8_t bases[32];
			for(int k = 0;k < 32;++k){ bases[k] = (uint8_t)q.at(lo - 32 + k); }
			m.seed_after_regular(bases);
		}
		for(uint64_t i = lo;i <= hi;++i){ m.push(q.at(i), int(i + 1), true); }
		have_state = true;
		state_next = hi + 1;
	}
	if(!(have_state && state_next == n_iter)){
		uint8_t bases[32];
		for(int k = 0;k < 32;++k){ bases[k] = (uint8_t)q.at(n_iter - 32 + k); }
		m.seed_after_regular(bases);
	}
	m.drain(int(n_iter + 1));

	// tie-break ordinals
	std::vector<size_t> idx(out.size() - first_out);
	for(size_t i = 0;i < idx.size();++i){ idx[i] = first_out + i; }
	std::stable_sort(idx.begin(), idx.end(), [&](size_t x, size_t y){
		if(out[x].loc != out[y].loc) return out[x].loc < out[y].loc;
		return out[x].strand < out[y].strand;
	});
	for(size_t i = 0;i < idx.size();){
		size_t j = i;
		while(j < idx.size() && out[idx[j]].loc == out[idx[i]].loc && out[idx[j]].strand == out[idx[i]].strand){ ++j; }
		if(j - i > 64){ return false; }
		for(size_t k = i;k < j;++k){ out[idx[k]].ord = (uint8_t)(k - i); }
		i = j;
	}
	return true;
}

// ---------------------------------------------------------------------------------------------
// Candidate oligo list of select_words (select_words.cpp:22-85): every assay oligo, plus all
// of its slot shifts towards the 5' end (optimize_5) and towards the 3' end (optimize_3), with
// the floor unsigned(size * threshold) (select_words.cpp:83: unsigned*float in float, truncated).
struct Candidate {
	Planes fwd;      // masks as stored (match against plus-strand windows and irregular words)
	Planes rc;       // reverse-complemented masks (match against plus-strand windows = minus-strand words)
	uint32_t floor_;
	uint32_t base;   // index of the unshifted candidate this one is a 5'/3' slot shift of (itself if unshifted)
	int32_t shift;   // slots `fwd` is shifted by relative to that candidate (towards slot 31 = positive); rc is shifted by -shift
};

inline void build_candidates(const uint64_t *pairs /* n x {F[2],R[2]} */, uint32_t n_pairs, bool opt5, bool opt3,
	float threshold, std::vector<Candidate> &out)
{
	auto emit = [&out, threshold](const Planes &w, uint32_t base, int32_t shift){
		Candidate c;
		c.fwd = w;
		c.rc = planes_revcomp(w);
		c.floor_ = (unsigned)((float)(unsigned)planes_size(w)*threshold);
		c.base = base; c.shift = shift;
		out.push_back(c);
	};
	out.reserve(out.size() + 2*(size_t)n_pairs*((opt5 || opt3) ? 8 : 1));
	for(uint32_t i = 0;i < n_pairs;++i){
		for(int o = 0;o < 2;++o){
			const Planes base = planes_of_word(pairs + 4*i + 2*o);
			const uint32_t bi = (uint32_t)out.size();
			emit(base, bi, 0);
			if(opt5 || opt3){
				const int cs = planes_start(base), ce = planes_stop(base);
				if(opt5 && cs > 0){ Planes t = base; for(int j = 0;j < cs;++j){ t = planes_shift_left(t); emit(t, bi, -(j + 1)); } }
				if(opt3 && ce < 31){ Planes t = base; for(int j = ce;j < 31;++j){ t = planes_shift_right(t); emit(t, bi, j - ce + 1); } }
			}
		}
	}
}

// ---------------------------------------------------------------------------------------------
// IUPAC expansion of an oligo in the order of Word::begin()/next() (word.h:525-647): an odometer
// whose fastest digit is slot 15, then 14 ... 0, then 31 ... 16 (the reference walks its two
// 64-bit blocks from the least significant nibble up); within a slot A -> C -> G -> T restricted
// to the slot's base set.  Each expansion is returned as base indices 0..3 (A,C,G,T) of the
// occupied slots start()..stop(), i.e. what Word::str() spells.
template<class F>
inline bool for_each_expansion(const Planes &w, size_t cap, F f /* (const uint8_t *bases, int len) */)
{
	int order[32];
	for(int i = 0;i < 16;++i){ order[i] = 15 - i; order[16 + i] = 31 - i; }
	uint8_t set[32], cur[32];
	for(int k = 0;k < 32;++k){ set[k] = (uint8_t)planes_nibble(w, k); cur[k] = set[k] ? (uint8_t)(set[k] & (0u - set[k])) : 0; }
	const int first = planes_start(w), last = planes_stop(w);
	if(last < first) return true;
	size_t n = 0;
	uint8_t s[32];
	while(true){
		if(n >= cap) return false;
		for(int k = first;k <= last;++k){
			const uint8_t c = cur[k];
			s[k - first] = (c == 1) ? 0 : (c == 2) ? 1 : (c == 4) ? 2 : (c == 8) ? 3 : 255;   // EOS inside an oligo: illegal base
		}
		f(s, last - first + 1); ++n;
		bool advanced = false;
		for(int oi = 0;oi < 32 && !advanced;++oi){
			const int k = order[oi];
			if(cur[k] == 0) continue;
			bool wrapped = false;
			unsigned c = cur[k];
			do{
				if(c == 8){ c = 1; wrapped = true; }
				else c <<= 1;
			} while(!(c & set[k]));
			cur[k] = (uint8_t)c;
			if(!wrapped) advanced = true;
		}
		if(!advanced) break;
	}
	return true;
}

inline bool expand_oligo(const Planes &w, std::vector<std::vector<uint8_t> > &out, size_t cap)
{
	return for_each_expansion(w, cap > out.size() ? cap - out.size() : 0, [&out](const uint8_t *b, int len){ out.push_back(std::vector<uint8_t>(b, b + len)); });
}

// Word::max_overlap (word.h:38-91): the DP there gives a cell its diagonal predecessor's count plus one where
// the two nibbles are equal, so the maximum is the best ungapped diagonal: for every relative shift, the
// number of slots (inside both words' start..stop ranges) holding the same code, over the longer word's size.
inline float max_overlap(const Planes &a, const Planes &b)
{
	const uint32_t oa = planes_occupied(a), ob = planes_occupied(b);
	if(!oa || !ob) return 0.0f;                                                  // an empty word is undefined in the reference (its unsigned loop bounds wrap); defined as 0 here
	auto range = [](uint32_t o){ const int lo = __builtin_ctz(o), hi = 31 - __builtin_clz(o); return (hi - lo == 31) ? 0xFFFFFFFFu : (((1u << (hi - lo + 1)) - 1u) << lo); };
	const uint32_t ra = range(oa), rb = range(ob);
	int best = 0;
	for(int d = -31;d <= 31;++d){                                                // b slot j meets a slot j + d
		auto sh = [d](uint32_t v){ return (d >= 0) ? (v << d) : (v >> (-d)); };
		const uint32_t both = ra & sh(rb);
		if(!both) continue;
		const uint32_t diff = (a.a ^ sh(b.a)) | (a.c ^ sh(b.c)) | (a.g ^ sh(b.g)) | (a.t ^ sh(b.t));
		best = std::max(best, __builtin_popcount(both & ~diff));
	}
	return float(best)/float(std::max(planes_size(a), planes_size(b)));
}

// PCR::compute_oligo_overlap (pcr_assay.cpp:736-754) with MULTIPLEX_OLIGO_REUSE_BONUS (assay.h:19)
inline float oligo_overlap(const Planes &f, const Planes &r, const Planes *pool_fr, uint32_t n_pool)
{
	float best_f = 0.0f, best_r = 0.0f;
	for(uint32_t i = 0;i < n_pool;++i){
		const Planes &pf = pool_fr[2*i], &pr = pool_fr[2*i + 1];
		best_f = std::max(best_f, max_overlap(f, pf)); best_f = std::max(best_f, max_overlap(f, pr));
		best_r = std::max(best_r, max_overlap(r, pf)); best_r = std::max(best_r, max_overlap(r, pr));
	}
	return ((best_f == 1.0f) ? 10.0f : best_f) + ((best_r == 1.0f) ? 10.0f : best_r);
}

// glibc rand_r (stdlib/rand_r.c, glibc 2.35): the reference's only random source (sample.cpp:12,
// pcr_assay.cpp:618-638, main.cpp:542).  Restated so that the sampler does not depend on the host libc.
inline uint32_t rand_r_glibc(uint32_t *seed)
{
	uint32_t next = *seed;
	next = next*1103515245u + 12345u;
	uint32_t result = (next/65536u) % 2048u;
	next = next*1103515245u + 12345u;
	result = (result << 10) ^ ((next/65536u) % 1024u);
	next = next*1103515245u + 12345u;
	result = (result << 10) ^ ((next/65536u) % 1024u);
	*seed = next;
	return result;
}

// Word::degeneracy() (word.h:97-138): product of the slot multiplicities, in double
inline double planes_degeneracy(const Planes &w)
{
	double r = 1.0;
	for(int k = 0;k < 32;++k){ const int d = __builtin_popcount(planes_nibble(w, k)); if(d) r *= d; }
	return r;
}

// ---------------------------------------------------------------------------------------------
// Seeds for the match scan (generalised pigeonhole).
//
// An orientation with m occupied slots and floor f tolerates k = m - f mismatching slots.  Cut the
// occupied slots into disjoint blocks and give block i a budget t_i in {0, 1, 2} with
// sum(t_i + 1) >= k + 1: if every block had more than t_i mismatches the window would have at least
// k + 1, so a window reaching the floor has SOME block with at most t_i mismatching slots.  The
// same holds for any sub-window of a block.  Every block therefore yields a set of 8-gram codes
// (2 bits per base) at a fixed slot offset:
//   t = 0, block of q >= 5 slots: the codes whose block positions lie in the oligo's base sets;
//          when q < 8 the 8-window is padded with don't-care positions (all 4 bases enumerated);
//   t = 1, 8 or 7 slots: those codes plus the ones with exactly one position outside its base set
//          (25 codes for a plain 8-mer, 4 x 22 for a padded 7-mer);
//   t = 2, 8 slots: plus the ones with exactly two positions outside (277 codes for a plain 8-mer).
// The scan looks the TARGET's 8-gram (9-gram in its second form) up at every position and evaluates exactly only the windows a
// hit implies.  Among the admissible block structures the cheapest is
// taken, cost = expected hits per target position = codes / 4^8.  A target 8-gram holding an IUPAC
// code can match seeds it is not equal to, so tiles containing such bases are scanned by the
// bit-sliced kernel instead (pcr_device.hip); so are orientations for which no structure exists
// (more than MAX_SEED_CODES codes: very low thresholds, or IUPAC slots that expand too far).
struct Seed { uint32_t code; uint16_t orient; uint8_t q; uint8_t off; };   // q = gram length (8, or 9 for the second form of the seed scan); orient = 2*candidate + {0: fwd, 1: rc}; off = slot of the window's first base

enum { SEED_Q = 8, SEED_Q_MAX = 9, MIN_SEED_BLOCK = 5, MAX_SEED_CODES = 512 /* per orientation */ };

// all codes whose position j lies in sets[j] (4-bit base sets, bit 0 = A ... bit 3 = T; first base in the LOW bits)
inline void seed_emit(const unsigned *sets, int Q, uint32_t orient, uint32_t off, std::vector<Seed> &out)
{
	uint32_t fixed = 0; unsigned total = 1;
	int nf = 0, fpos[SEED_Q_MAX]; unsigned nd[SEED_Q_MAX]; uint8_t base[SEED_Q_MAX][4];
	for(int j = 0;j < Q;++j){
		const unsigned st = sets[j];
		if(st == 0) return;
		if((st & (st - 1)) == 0){ fixed |= (uint32_t)__builtin_ctz(st) << (2*j); continue; }
		unsigned n = 0;
		for(unsigned b = 0;b < 4;++b){ if(st & (1u << b)) base[nf][n++] = (uint8_t)b; }
		nd[nf] = n; fpos[nf] = j; ++nf; total *= n;
	}
	const size_t at = out.size();
	out.resize(at + total);
	Seed *dst = out.data() + at;
	Seed sd; sd.orient = (uint16_t)orient; sd.q = (uint8_t)Q; sd.off = (uint8_t)off;
	for(unsigned idx = 0;idx < total;++idx){
		uint32_t code = fixed; unsigned r = idx;
		for(int f = 0;f < nf;++f){ code |= (uint32_t)base[f][r % nd[f]] << (2*fpos[f]); r /= nd[f]; }
		sd.code = code;
		dst[idx] = sd;
	}
}

inline unsigned seed_count(const unsigned *sets, int Q)
{
	unsigned n = 1;
	for(int j = 0;j < Q;++j) n *= (unsigned)__builtin_popcount(sets[j]);
	return n;
}

// codes a block contributes: every window position inside its base set, or up to `budget` positions outside theirs
// (don't-care positions, sets[j] = 15, have no outside)
inline unsigned seed_block_codes(const unsigned *sets, int Q, int budget)
{
	const unsigned c0 = seed_count(sets, Q);
	if(c0 == 0 || budget == 0) return c0;
	unsigned d[SEED_Q_MAX];
	for(int j = 0;j < Q;++j) d[j] = (unsigned)__builtin_popcount(sets[j]);
	unsigned c = c0;
	for(int j = 0;j < Q;++j){
		if(d[j] == 4) continue;
		const unsigned cj = c0/d[j]*(4 - d[j]);
		c += cj;
		if(budget >= 2){ for(int i = j + 1;i < Q;++i){ if(d[i] < 4) c += cj/d[i]*(4 - d[i]); } }
	}
	return c;
}

inline void seed_emit_block(unsigned *sets, int Q, uint32_t orient, uint32_t off, int budget, std::vector<Seed> &out)
{
	seed_emit(sets, Q, orient, off, out);                                  // no mismatch in the window
	if(budget < 1) return;
	for(int j = 0;j < Q;++j){                                              // exactly one, at position j
		const unsigned keep = sets[j], outside = ~keep & 15u;
		if(!outside) continue;
		sets[j] = outside;
		seed_emit(sets, Q, orient, off, out);
		if(budget >= 2){                                                   // exactly two, at positions j < i
			for(int i = j + 1;i < Q;++i){
				const unsigned keep_i = sets[i], outside_i = ~keep_i & 15u;
				if(!outside_i) continue;
				sets[i] = outside_i;
				seed_emit(sets, Q, orient, off, out);
				sets[i] = keep_i;
			}
		}
		sets[j] = keep;
	}
}

// A block structure: n0 blocks with budget 0 sharing what the others leave as evenly as possible, then n1s blocks of Q - 1 slots
// and n1 - n1s of Q slots with budget 1, then n2 blocks of Q slots with budget 2; n0 + 2*n1 + 3*n2 >= k + 1.  Blocks
// shorter than Q are padded to a Q-window with don't-care positions.  each() lays the blocks out, calling f(pos, len, budget).
struct SeedLayout {
	int n0, n1, n1s, n2;
	template<class F> void each(int first, int size, int Q, F f) const
	{
		const int rest = size - Q*(n1 - n1s) - (Q - 1)*n1s - Q*n2;
		int pos = first;
		for(int b = 0;b < n0;++b){ const int len = rest/n0 + ((b < rest % n0) ? 1 : 0); f(pos, std::min(len, Q), 0); pos += len; }
		for(int b = 0;b < n1s;++b){ f(pos, Q - 1, 1); pos += Q - 1; }
		for(int b = 0;b < n1 - n1s;++b){ f(pos, Q, 1); pos += Q; }
		for(int b = 0;b < n2;++b){ f(pos, Q, 2); pos += Q; }
	}
};

// Appends the seeds of one orientation; returns false (nothing appended) if it cannot be seeded.
// max_exact_pos (optional): largest first slot of a padded block (-1 if none); while it stays <= 32 - Q under a
// slot shift of the oligo the seeds of the shifted oligo are these seeds with `off` moved by the shift.
// Q: gram length (8 bases = 16-bit codes, the first form of the scan; 9 = 18-bit codes, the second form: three times fewer
// false seed hits per target position for the same oligo, since the code space grows faster than the code lists).
inline bool orientation_seeds(const Planes &m, uint32_t floor_, uint32_t orient, std::vector<Seed> &out, int *max_exact_pos = nullptr, int Q = SEED_Q)
{
	if(max_exact_pos) *max_exact_pos = -1;
	const uint32_t occ = m.a | m.c | m.g | m.t;
	const int size = __builtin_popcount(occ);
	if(size == 0 || floor_ == 0 || floor_ > (uint32_t)size) return floor_ > (uint32_t)size;   // dead orientation: trivially "seeded" with no seeds
	const int k = size - (int)floor_;
	const int first = __builtin_ctz(occ);
	if((occ >> first) != ((size == 32) ? 0xFFFFFFFFu : ((1u << size) - 1u))) return false;   // holes: leave it to the bit-sliced scan
	unsigned slot_set[32];
	for(int j = 0;j < first;++j) slot_set[j] = 0;
	for(int j = first;j < first + size;++j) slot_set[j] = planes_nibble(m, j);
	for(int j = first + size;j < 32;++j) slot_set[j] = 0;

	auto window = [&](int pos, int len, unsigned *sets) -> int {           // the Q-window holding slots [pos, pos + len)
		const int ws = std::min(pos, 32 - Q);
		for(int j = 0;j < Q;++j){ const int sl = ws + j; sets[j] = (sl >= pos && sl < pos + len) ? slot_set[sl] : 15u; }
		return ws;
	};
	SeedLayout best = { 0, 0, 0, 0 }; bool have = false; unsigned best_codes = 0;
	for(int n2 = 0;3*(n2 - 1) < k + 1;++n2){
		for(int n1 = 0;3*n2 + 2*(n1 - 1) < k + 1;++n1){
			const int n0 = std::max(0, k + 1 - 3*n2 - 2*n1);
			if(n0 + n1 + n2 == 0) continue;
			for(int n1s = 0;n1s <= n1;++n1s){
				const int rest = size - Q*(n1 - n1s) - (Q - 1)*n1s - Q*n2;
				if(rest < 0 || (n0 > 0 && rest/n0 < MIN_SEED_BLOCK)) continue;
				const SeedLayout lay = { n0, n1, n1s, n2 };
				unsigned codes = 0;
				lay.each(first, size, Q, [&](int pos, int len, int budget){
					unsigned sets[SEED_Q_MAX];
					window(pos, len, sets);
					if(codes <= MAX_SEED_CODES) codes += seed_block_codes(sets, Q, budget);
				});
				if(codes > MAX_SEED_CODES) continue;
				if(!have || codes < best_codes){ have = true; best = lay; best_codes = codes; }   // cost = expected seed hits per target position
			}
		}
	}
	if(!have) return false;
	best.each(first, size, Q, [&](int pos, int len, int budget){
		unsigned sets[SEED_Q_MAX];
		const int ws = window(pos, len, sets);
		seed_emit_block(sets, Q, orient, (uint32_t)ws, budget, out);
		if(len < Q && max_exact_pos) *max_exact_pos = std::max(*max_exact_pos, pos);
	});
	return true;
}

// ---------------------------------------------------------------------------------------------
// Trial words of the local-search moves (optimize_pcr.cpp), in the reference's order, after its
// degeneracy / length gates and before is_valid:
//   0 increase_degeneracy (:17-19 gate, :54-76): every occupied slot, every base bit not yet set,
//                          skipped if the trial's degeneracy exceeds max_degen;
//   1 decrease_degeneracy (:232-247): every occupied slot, every set bit whose removal leaves a
//                          non-empty, different set;
//   2 trim5 / 3 trim3     (:391-399): drop the first / last occupied slot unless at primer_min;
//   4 grow5 / 5 grow3     (:671-673, :709-713): A, C, G, T in the slot before the first / after the
//                          last (Word::grow_front/back leave the word unchanged if there is no room)
//                          unless at primer_max.
// Slots are not re-centred (the reference centres only the accepted word, optimize.cpp:152).
inline bool move_trials(const Planes &cur, int move, double max_degen, int primer_min, int primer_max, std::vector<Planes> &out)
{
	const int first = planes_start(cur), last = planes_stop(cur), len = planes_size(cur);
	auto with_nibble = [](Planes w, int k, unsigned v){
		const uint32_t bit = 1u << k;
		w.a = (w.a & ~bit) | ((v & 1u) ? bit : 0u); w.c = (w.c & ~bit) | ((v & 2u) ? bit : 0u);
		w.g = (w.g & ~bit) | ((v & 4u) ? bit : 0u); w.t = (w.t & ~bit) | ((v & 8u) ? bit : 0u);
		return w;
	};
	switch(move){
		case 0:
			if(planes_degeneracy(cur) >= max_degen) return true;
			for(int i = first;i <= last;++i){
				const unsigned c = planes_nibble(cur, i);
				for(unsigned b = 1;b <= 8;b <<= 1){
					if(c & b) continue;
					const Planes w = with_nibble(cur, i, c | b);
					if(planes_degeneracy(w) > max_degen) continue;
					out.push_back(w);
				}
			}
			return true;
		case 1:
			for(int i = first;i <= last;++i){
				const unsigned c = planes_nibble(cur, i);
				for(unsigned b = 1;b <= 8;b <<= 1){
					const unsigned d = c & ~b;
					if(!d || d == c) continue;
					out.push_back(with_nibble(cur, i, d));
				}
			}
			return true;
		case 2: if(len != primer_min){ out.push_back(first < 32 ? with_nibble(cur, first, 0) : cur); } return true;
		case 3: if(len != primer_min){ out.push_back(last >= 0 ? with_nibble(cur, last, 0) : cur); } return true;
		case 4:
			if(len == primer_max) return true;
			for(unsigned b = 1;b <= 8;b <<= 1) out.push_back(first - 1 >= 0 && first < 32 ? with_nibble(cur, first - 1, b) : cur);
			return true;
		case 5:
			if(len == primer_max) return true;
			for(unsigned b = 1;b <= 8;b <<= 1) out.push_back(last + 1 < 32 ? with_nibble(cur, last + 1, b) : cur);
			return true;
		default: return false;
	}
}

} // namespace pcrhost
#endif
