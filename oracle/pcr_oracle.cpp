// TEST INFRASTRUCTURE ONLY -- see pcr_oracle.h.  Plain scalar C++; no SIMD, no threads.
// Every block cites the reference (LANL-Bioinformatics/PCRamp v0.3) file:line it restates.
#include "pcr_oracle.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <deque>
#include <map>
#include <set>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace {

enum { EOS = 0, BA = 1, BC = 2, BG = 4, BT = 8 };
enum { PLUS = 1, MINUS = 2 };

// ------------------------------------------------------------------------------------ Word
// 32 slots, one IUPAC nibble each; slot 0 is the 5' end (word.cpp:11-16).
struct W {
	uint64_t b[2];
	W() { b[0] = b[1] = 0; }
	bool operator==(const W &r) const { return b[0] == r.b[0] && b[1] == r.b[1]; }
	bool operator<(const W &r) const { return (b[0] != r.b[0]) ? (b[0] < r.b[0]) : (b[1] < r.b[1]); } // word.h:197
	unsigned get(int k) const { return (unsigned)(b[k >> 4] >> ((15 - (k & 15))*4)) & 0xF; }       // word.h:290
	void set(unsigned v, int k) {                                                                  // word.h:307
		const int sh = (15 - (k & 15))*4;
		b[k >> 4] = (b[k >> 4] & ~(uint64_t(0xF) << sh)) | (uint64_t(v) << sh);
	}
	int start() const { for(int k = 0;k < 32;++k){ if(get(k)) return k; } return 32; }            // word.h:256
	int stop() const { for(int k = 31;k >= 0;--k){ if(get(k)) return k; } return -1; }            // word.h:273
	unsigned size() const { unsigned n = 0; for(int k = 0;k < 32;++k){ n += (get(k) != 0); } return n; } // word.cpp:199
	void shift_left() { b[0] = (b[0] << 4) | (b[1] >> 60); b[1] <<= 4; }                          // word.cpp:215
	void shift_right() { b[1] = (b[1] >> 4) | (b[0] << 60); b[0] >>= 4; }                         // word.cpp:224
	void push_back(unsigned v) {                                                                   // word.cpp:31-48
		const int last = stop() + 1;
		if(last < 32){ set(v, last); return; }
		shift_left();
		b[1] |= v;
	}
	double degeneracy() const {                                                                    // word.h:97-138
		double r = 1.0;
		for(int k = 0;k < 32;++k){
			const unsigned d = __builtin_popcount(get(k));
			if(d){ r *= d; }
		}
		return r;
	}
	W complement() const {                                                                         // word.h:140-183
		W r;
		const int first = start(), last = stop();
		int dst = 0;
		for(int src = last;src >= first;--src, ++dst){
			const unsigned v = get(src);
			unsigned c = 0;
			if(v & BA) c |= BT;
			if(v & BT) c |= BA;
			if(v & BG) c |= BC;
			if(v & BC) c |= BG;
			r.set(c, dst);
		}
		return r;
	}
	void center() {                                                                                // word.h:392-418
		int left = start();
		int right = stop();
		if(left > right) return;
		right = 32 - right;
		const int delta = (right - left)/2;   // C++ truncation toward zero
		if(delta > 0){ for(int i = 0;i < delta;++i) shift_right(); }
		else{ for(int i = 0;i > delta;--i) shift_left(); }
	}
};

// # of slots whose base sets intersect (word.cpp:68-154).
inline unsigned match_count(const W &x, const W &y)
{
	unsigned n = 0;
	for(int h = 0;h < 2;++h){
		uint64_t a = x.b[h] & y.b[h];
		a = (a | (a >> 1) | (a >> 2) | (a >> 3)) & 0x1111111111111111ULL;
		n += (unsigned)__builtin_popcountll(a);
	}
	return n;
}

inline unsigned base_bits(char c, bool &ok)                                                        // base_table.h:30-76
{
	ok = true;
	switch(c){
		case 'A': case 'a': return 1;
		case 'C': case 'c': return 2;
		case 'G': case 'g': return 4;
		case 'T': case 't': case 'U': case 'u': return 8;
		case 'M': case 'm': return 1|2;
		case 'R': case 'r': return 4|1;
		case 'S': case 's': return 4|2;
		case 'V': case 'v': return 4|2|1;
		case 'W': case 'w': return 1|8;
		case 'Y': case 'y': return 8|2;
		case 'H': case 'h': return 1|2|8;
		case 'K': case 'k': return 4|8;
		case 'D': case 'd': return 4|1|8;
		case 'B': case 'b': return 4|8|2;
		case 'N': case 'n': case 'I': case 'i': case 'X': case 'x': return 15;
		case '-': return 0;
		default: ok = false; return 0;
	}
}

inline bool is_degen(unsigned v) { return !(v == 1 || v == 2 || v == 4 || v == 8); }              // base_table.h:125-139

// word.cpp:233-294.  Table 2 of Li et al., Genomics 83 (2004) 311-320; rows = template pair,
// columns = primer pair, both ordered {CC,GC,AC,TC,CG,GG,AG,TG,CA,GA,AA,TA,CT,GT,AT,TT}.
const float TAQ_MAMA[256] = {
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

inline int taq_index(unsigned v)                                                                   // word.cpp:233-247
{
	switch(v){ case BC: return 0; case BG: return 1; case BA: return 2; case BT: return 3; }
	return -1;
}

float taq_mama(unsigned p1, unsigned p2, unsigned t1, unsigned t2)                                  // word.cpp:249-294
{
	const int a = taq_index(p1), b = taq_index(p2), c = taq_index(t1), d = taq_index(t2);
	if(a < 0 || b < 0 || c < 0 || d < 0) return 1.0f;
	return std::min(1.0f, TAQ_MAMA[16*(4*d + c) + (4*b + a)]);
}

// ------------------------------------------------------------------------------------ Sequence
struct Seq {
	std::vector<uint8_t> buf;   // high nibble first (sequence.h:223-228)
	uint64_t len = 0;
	float weight = 1.0f;
	bool active = true;
	unsigned at(uint64_t i) const { const uint8_t v = buf[i >> 1]; return (i & 1) ? (v & 0xF) : (v >> 4); }
	void split(uint64_t i) { uint8_t &v = buf[i >> 1]; v = (i & 1) ? (v & 0xF0) : (v & 0x0F); } // sequence.h:228-241
};

struct Entry { W w; int32_t loc; uint32_t index; uint32_t strand; };

// Sequence::pack, sequence.cpp:92-267.  The streaming window, the size counter that ignores
// EOS, the GC ring of the last 32 pushed nibbles, the centred partial words at both ends, the
// trailing shift-left loop -- all kept step for step.
void pack(const Seq &s, unsigned index, unsigned degen_thr, float min_gc, float max_gc,
	unsigned min_len, std::vector<Entry> &db)
{
	W w;
	size_t cws = 0;
	const bool gc_filter = (min_gc > 0.0f) || (max_gc < 1.0f);
	std::deque<uint8_t> gc;
	unsigned num_gc = 0;
	const float norm = 1.0f/32;
	const size_t nbytes = s.buf.size();
	int loc = 1;
	size_t it = 0;

	for(;it != nbytes;++loc){                                                    // :110
		unsigned b;
		if(loc % 2 == 1){ b = s.buf[it] >> 4; }
		else{ b = s.buf[it] & 0xF; ++it; }
		w.push_back(b);                                                           // :122
		cws += (b != EOS);                                                        // :125
		if(gc_filter){                                                            // :127-146
			if(gc.size() == 32){ num_gc -= ((gc.front() & (BG|BC)) != 0); gc.pop_front(); }
			gc.push_back((uint8_t)b);
			num_gc += ((b & (BG|BC)) != 0);
			const float frac = num_gc*norm;
			if(frac < min_gc || frac > max_gc){ cws = std::min(cws, size_t(31)); continue; }
		}
		if(w.degeneracy() > degen_thr){ cws = std::min(cws, size_t(31)); continue; } // :149-153
		if(cws < 32){                                                             // :155-179
			if(cws >= min_len){
				W t = w;
				t.center();
				db.push_back(Entry{t, loc - int(cws) - t.start(), index, PLUS});
				t = t.complement();
				t.center();
				db.push_back(Entry{t, loc - 1 + t.start(), index, MINUS});
			}
		}
		else{                                                                     // :181-194
			db.push_back(Entry{w, int(loc - cws), index, PLUS});
			db.push_back(Entry{w.complement(), loc - 1, index, MINUS});
			--cws;
		}
	}

	while(cws > 0){                                                               // :198-263
		w.shift_left();
		--cws;
		if(gc_filter){
			if(gc.size() == 32){ num_gc -= ((gc.front() & (BG|BC)) != 0); gc.pop_front(); }
			const float frac = num_gc*norm;
			if(frac < min_gc || frac > max_gc){ continue; }
		}
		if(w.degeneracy() > degen_thr){ continue; }
		if(cws >= min_len){
			W t = w;
			t.center();
			db.push_back(Entry{t, loc - 1 - int(cws) - t.start(), index, PLUS});
			t = t.complement();
			t.center();
			db.push_back(Entry{t, loc - 2 + t.start(), index, MINUS});
		}
	}
}

// Sequence::has_split, sequence.cpp:304-330 (throws on an out-of-range request).
bool has_split(const Seq &s, int loc, int len)
{
	if( ((int64_t)loc + len > (int64_t)s.len) || loc < 0 || len < 0 ){
		throw "has_split: range is out of bounds";
	}
	for(int i = 0;i < len;++i){ if(s.at(loc + i) == EOS) return true; }
	return false;
}

inline bool entry_key_less(const Entry &a, const Entry &b) { return a.w < b.w; }

} // namespace

// ------------------------------------------------------------------------------------ session
struct orc_session {
	std::vector<Seq> seq;
	std::vector<Entry> db;        // global target_db, sorted by key (read_only_multimap.h:93)
	std::vector<W> keys;          // keys(target_db) (pcramp.h:231-256)
	orc_options opt;
	std::string err;
	orc_session() {
		opt.target_threshold = 1.0f; opt.search_multiplier = 0.9f;
		opt.amp_min = 80; opt.amp_max = 200; opt.use_taq_mama = 0;
		opt.pack_max_degen = 256; opt.pack_min_gc = 0.0f; opt.pack_max_gc = 1.0f;
		opt.min_primer = 18; opt.optimize_5 = 0; opt.optimize_3 = 0;
	}
};

namespace {

// select_words, select_words.cpp:8-139: per candidate oligo (and its 5'/3' slot shifts), the
// arg-max-with-ties set of unique source words at or above unsigned(size*threshold); the union
// of those keys brings all of their occurrences in this sequence.
void select_words(std::vector<Entry> &dst, std::vector<Entry> &src, const std::vector<W> &oligos,
	bool opt5, bool opt3, float threshold)
{
	if(src.empty() || oligos.empty()) return;
	std::stable_sort(src.begin(), src.end(), entry_key_less);
	std::vector<W> skeys;                                                         // keys(m_src)
	for(const Entry &e : src){ if(skeys.empty() || !(skeys.back() == e.w)) skeys.push_back(e.w); }

	std::vector<W> cand;
	for(const W &o : oligos){                                                     // :25-75
		cand.push_back(o);
		if(opt5 || opt3){
			const int cs = o.start(), ce = o.stop();
			if(opt5 && cs > 0){ W t = o; for(int j = 0;j < cs;++j){ t.shift_left(); cand.push_back(t); } }
			if(opt3 && ce < 31){ W t = o; for(int j = ce;j < 31;++j){ t.shift_right(); cand.push_back(t); } }
		}
	}

	std::set<size_t> matched;
	for(const W &c : cand){
		unsigned best = (unsigned)(c.size()*threshold);                           // :83  (unsigned*float -> float -> unsigned)
		std::vector<size_t> buf;
		for(size_t j = 0;j < skeys.size();++j){                                   // :100-117
			const unsigned n = match_count(c, skeys[j]);
			if(n >= best){
				if(best < n) buf.clear();
				best = n;
				buf.push_back(j);
			}
		}
		matched.insert(buf.begin(), buf.end());
	}
	for(size_t k : matched){                                                      // :126-138
		Entry probe; probe.w = skeys[k];
		auto r = std::equal_range(src.begin(), src.end(), probe, entry_key_less);
		dst.insert(dst.end(), r.first, r.second);
	}
}

struct OligoMatch { int32_t loc; uint32_t index; uint32_t strand; uint32_t key; uint8_t o; };
struct Amplicon { uint32_t index; float weight; uint32_t f, r; uint8_t orient; };

// match_words, optimize.cpp:291-301
void match_words(std::vector<uint32_t> &out, const W &o, const std::vector<W> &keys, float thr)
{
	const unsigned scaled = (unsigned)(o.size()*thr);
	for(size_t k = 0;k < keys.size();++k){ if(match_count(o, keys[k]) >= scaled) out.push_back((uint32_t)k); }
}

// find_oligo_match, optimize.cpp:263-289
void find_oligo_match(std::vector<OligoMatch> &out, const std::vector<uint32_t> &wm, uint8_t oligo,
	uint32_t strand, const orc_session &s)
{
	for(uint32_t k : wm){
		Entry probe; probe.w = s.keys[k];
		auto r = std::equal_range(s.db.begin(), s.db.end(), probe, entry_key_less);
		for(auto j = r.first;j != r.second;++j){
			if(!(j->strand & strand)) continue;
			if(!s.seq[j->index].active) continue;
			out.push_back(OligoMatch{j->loc, j->index, j->strand, k, oligo});
		}
	}
}

inline int loc5(const OligoMatch &m, int start, int stop) { return (m.strand == PLUS) ? m.loc + start : m.loc - stop; }  // sequence.h:57-65
inline int loc3(const OligoMatch &m, int start, int stop) { return (m.strand == PLUS) ? m.loc + stop : m.loc - start; }  // sequence.h:67-75

// PCR::find_amplicon_match, pcr_assay.cpp:338-441
void find_amplicon_match(std::vector<Amplicon> &amp, const std::vector<OligoMatch> &m, uint8_t plus_o,
	uint8_t minus_o, const W &plus_w, const W &minus_w, const orc_session &s, int amp_min, int amp_max,
	uint8_t orient)
{
	const int ps = plus_w.start(), pe = plus_w.stop(), ms = minus_w.start(), me = minus_w.stop();
	for(size_t p = 0;p < m.size();++p){
		if(m[p].o != plus_o) continue;
		for(size_t q = p;q < m.size();++q){
			if(m[p].index != m[q].index) break;
			if(m[q].o != minus_o) continue;
			if(loc3(m[p], ps, pe) >= loc5(m[q], ms, me)) continue;
			int amp_start = loc5(m[p], ps, pe);
			const int amp_stop = std::min(loc3(m[q], ms, me), int(s.seq[m[p].index].len - 1));
			int amp_len = amp_stop - amp_start + 1;
			if(amp_len < amp_min) continue;
			if(amp_len > amp_max) break;
			if(amp_start < 0){ amp_len += amp_start; amp_start = 0; }
			if(has_split(s.seq[m[p].index], amp_start, amp_len)) break;
			if(plus_o == 0) amp.push_back(Amplicon{m[p].index, s.seq[m[p].index].weight, m[p].key, m[q].key, orient});
			else amp.push_back(Amplicon{m[p].index, s.seq[m[p].index].weight, m[q].key, m[p].key, orient});
		}
	}
}

inline bool om_less(const OligoMatch &a, const OligoMatch &b)                     // assay.h:48-61
{
	if(a.index != b.index) return a.index < b.index;
	return a.loc < b.loc;
}

// PCR::collect_candidates, pcr_assay.cpp:12-69
void collect_candidates(std::vector<Amplicon> &amp, std::map<uint32_t, float> &fi, std::map<uint32_t, float> &ri,
	const W &F, const W &R, const orc_session &s, float thr, int amp_min, int amp_max)
{
	amp.clear();
	std::vector<uint32_t> fm, rm;
	match_words(fm, F, s.keys, thr*thr);                                          // :31-32
	match_words(rm, R, s.keys, thr*thr);
	std::vector<OligoMatch> om;
	find_oligo_match(om, fm, 0, PLUS, s);
	find_oligo_match(om, rm, 1, MINUS, s);
	std::stable_sort(om.begin(), om.end(), om_less);
	find_amplicon_match(amp, om, 0, 1, F, R, s, amp_min, amp_max, 1);
	om.clear();
	find_oligo_match(om, fm, 0, MINUS, s);
	find_oligo_match(om, rm, 1, PLUS, s);
	std::stable_sort(om.begin(), om.end(), om_less);
	find_amplicon_match(amp, om, 1, 0, R, F, s, amp_min, amp_max, 2);
	fi.clear(); ri.clear();
	for(const Amplicon &a : amp){ fi[a.f] = 0.0f; ri[a.r] = 0.0f; }
}

// update_identity, optimize.cpp:209-261
void update_identity(std::map<uint32_t, float> &ident, const W &w, const std::vector<W> &keys, bool use_taq)
{
	if(ident.empty()) return;
	const unsigned len = w.size();
	const float norm = 1.0/len;                                                   // double divide, rounded to float (:221)
	for(auto &kv : ident){ kv.second = match_count(w, keys[kv.first])*norm; }
	if(!use_taq) return;
	const int last = w.stop(), pen = last - 1;
	const unsigned p1 = w.get(pen), p2 = w.get(last);
	if(!is_degen(p1) && !is_degen(p2)){
		for(auto &kv : ident){
			const W &k = keys[kv.first];
			const unsigned t1 = k.get(pen), t2 = k.get(last);
			if(!is_degen(t1) && !is_degen(t2)){ kv.second *= taq_mama(p1, p2, t1, t2); }
		}
	}
}

W load_word(const uint64_t *p) { W w; w.b[0] = p[0]; w.b[1] = p[1]; return w; }

} // namespace

extern "C" {

unsigned orc_word_and(const uint64_t a[2], const uint64_t b[2]) { return match_count(load_word(a), load_word(b)); }
unsigned orc_word_size(const uint64_t a[2]) { return load_word(a).size(); }
int orc_word_start(const uint64_t a[2]) { return load_word(a).start(); }
int orc_word_stop(const uint64_t a[2]) { return load_word(a).stop(); }
double orc_word_degeneracy(const uint64_t a[2]) { return load_word(a).degeneracy(); }

int orc_word_from_string(const char *s, uint64_t out[2])                          // word.h:234-249
{
	const size_t n = strlen(s);
	if(n > 32) return -1;
	W w;
	for(size_t i = 0;i < n;++i){
		bool ok;
		const unsigned v = base_bits(s[i], ok);
		if(!ok) return -1;
		w.set(v, (int)i);
	}
	out[0] = w.b[0]; out[1] = w.b[1];
	return 0;
}

void orc_word_center(uint64_t w[2]) { W x = load_word(w); x.center(); w[0] = x.b[0]; w[1] = x.b[1]; }
void orc_word_complement(const uint64_t in[2], uint64_t out[2]) { W x = load_word(in).complement(); out[0] = x.b[0]; out[1] = x.b[1]; }
void orc_word_shift_left(uint64_t w[2]) { W x = load_word(w); x.shift_left(); w[0] = x.b[0]; w[1] = x.b[1]; }
void orc_word_shift_right(uint64_t w[2]) { W x = load_word(w); x.shift_right(); w[0] = x.b[0]; w[1] = x.b[1]; }

// Word::begin / Word::next, word.h:525-647: an odometer over the IUPAC expansions.  The
// reference walks bytes from the least-significant byte of buffer[0] upward, low nibble before
// high nibble, i.e. slot order 15,14,...,0,31,30,...,16; within a slot A -> C -> G -> T (lowest
// set bit first), skipping bases outside the slot's set.
int orc_word_expand(const uint64_t in[2], uint64_t *out, int cap)
{
	const W src = load_word(in);
	int order[32];
	for(int i = 0;i < 16;++i){ order[i] = 15 - i; order[16 + i] = 31 - i; }
	W it;
	for(int k = 0;k < 32;++k){
		const unsigned v = src.get(k);
		if(v){ it.set(v & (0u - v), k); }   // lowest set bit
	}
	int n = 0;
	while(true){
		if(n < cap){ out[2*n] = it.b[0]; out[2*n + 1] = it.b[1]; }
		++n;
		bool advanced = false;
		for(int oi = 0;oi < 32 && !advanced;++oi){
			const int k = order[oi];
			unsigned cur = it.get(k);
			if(cur == EOS) continue;
			const unsigned set = src.get(k);
			bool wrapped = false;
			do{
				if(cur == BT){ cur = BA; wrapped = true; }
				else{ cur <<= 1; }
			} while(!(cur & set));
			it.set(cur, k);
			if(!wrapped) advanced = true;
		}
		if(!advanced) break;
	}
	return n;
}

float orc_taq_mama(unsigned p1, unsigned p2, unsigned t1, unsigned t2) { return taq_mama(p1, p2, t1, t2); }

static bool seq_from_text(Seq &q, const char *seq)                                // sequence.cpp:11-39
{
	const size_t n = strlen(seq);
	q.len = n;
	q.buf.assign((n + 1)/2, 0);
	for(size_t i = 0;i < n;++i){
		bool ok;
		const unsigned v = base_bits(seq[i], ok);
		if(!ok) return false;
		q.buf[i >> 1] |= (i & 1) ? v : (v << 4);
	}
	return true;
}

long orc_pack(const char *seq, unsigned index, unsigned degen_thr, float min_gc, float max_gc,
	unsigned min_len, orc_entry *out, long cap)
{
	Seq q;
	if(!seq_from_text(q, seq)) return -1;
	std::vector<Entry> db;
	pack(q, index, degen_thr, min_gc, max_gc, min_len, db);
	std::stable_sort(db.begin(), db.end(), entry_key_less);
	long n = 0;
	for(const Entry &e : db){
		if(n < cap){
			out[n].w[0] = e.w.b[0]; out[n].w[1] = e.w.b[1];
			out[n].loc = e.loc; out[n].index = e.index; out[n].strand = e.strand; out[n].pad = 0;
		}
		++n;
	}
	return n;
}

orc_session *orc_session_create(void) { return new orc_session(); }
void orc_session_destroy(orc_session *s) { delete s; }
const char *orc_session_error(orc_session *s) { return s->err.c_str(); }

int orc_session_add_target(orc_session *s, const char *seq, float weight, int active)
{
	Seq q;
	if(!seq_from_text(q, seq)){ s->err = "illegal base"; return -1; }
	q.weight = weight; q.active = (active != 0);
	s->seq.push_back(q);
	return 0;
}

int orc_session_add_target_packed(orc_session *s, const uint8_t *packed, uint64_t len, float weight, int active)
{
	Seq q;
	q.len = len;
	q.buf.assign(packed, packed + (len + 1)/2);
	q.weight = weight; q.active = (active != 0);
	s->seq.push_back(q);
	return 0;
}

int orc_session_set_active(orc_session *s, unsigned idx, int active)
{
	if(idx >= s->seq.size()) return -1;
	s->seq[idx].active = (active != 0);
	return 0;
}

int orc_session_split(orc_session *s, unsigned idx, unsigned pos)
{
	if(idx >= s->seq.size() || pos >= s->seq[idx].len) return -1;
	s->seq[idx].split(pos);
	return 0;
}

void orc_session_set_options(orc_session *s, const orc_options *o) { s->opt = *o; }

long orc_session_select(orc_session *s, const uint64_t *pairs, unsigned n_pairs, float threshold, int min_len_override)
{
	std::vector<W> oligos;
	for(unsigned i = 0;i < n_pairs;++i){ oligos.push_back(load_word(pairs + 4*i)); oligos.push_back(load_word(pairs + 4*i + 2)); }
	const orc_options &o = s->opt;
	const float thr = (threshold < 0.0f) ? o.target_threshold*o.search_multiplier : threshold;   // main.cpp:669
	const unsigned min_len = (min_len_override >= 0) ? (unsigned)min_len_override : (unsigned)std::max(0, o.min_primer);
	s->db.clear();
	for(size_t i = 0;i < s->seq.size();++i){                                      // main.cpp:644-676
		if(!s->seq[i].active) continue;
		std::vector<Entry> local;
		pack(s->seq[i], (unsigned)i, o.pack_max_degen, o.pack_min_gc, o.pack_max_gc, min_len, local);
		select_words(s->db, local, oligos, o.optimize_5 != 0, o.optimize_3 != 0, thr);
	}
	std::stable_sort(s->db.begin(), s->db.end(), entry_key_less);                 // main.cpp:679
	s->keys.clear();                                                              // main.cpp:691
	for(const Entry &e : s->db){ if(s->keys.empty() || !(s->keys.back() == e.w)) s->keys.push_back(e.w); }
	return (long)s->db.size();
}

long orc_session_db_entries(orc_session *s, orc_entry *out, long cap)
{
	long n = 0;
	for(const Entry &e : s->db){
		if(n < cap){
			out[n].w[0] = e.w.b[0]; out[n].w[1] = e.w.b[1];
			out[n].loc = e.loc; out[n].index = e.index; out[n].strand = e.strand; out[n].pad = 0;
		}
		++n;
	}
	return n;
}

int orc_session_target_match(orc_session *s, const uint64_t pair[4], unsigned char *bits_out, unsigned char *orient_out)
{
	try{
		const W F = load_word(pair), R = load_word(pair + 2);
		const size_t T = s->seq.size();
		memset(bits_out, 0, T);
		if(orient_out) memset(orient_out, 0, T);
		std::vector<Amplicon> amp;
		std::map<uint32_t, float> fi, ri;
		collect_candidates(amp, fi, ri, F, R, *s, s->opt.target_threshold, s->opt.amp_min, s->opt.amp_max); // :554-557
		if(amp.empty()) return 0;
		update_identity(fi, F, s->keys, s->opt.use_taq_mama != 0);
		update_identity(ri, R, s->keys, s->opt.use_taq_mama != 0);
		for(const Amplicon &a : amp){                                             // :565-577
			const float local = sqrtf(fi[a.f]*ri[a.r]);
			if(local >= s->opt.target_threshold){
				bits_out[a.index] = 1;
				if(orient_out) orient_out[a.index] |= a.orient;
			}
		}
		return 0;
	}
	catch(const char *e){ s->err = e; return -1; }
}

float orc_session_target_coverage(orc_session *s, const uint64_t pair[4])
{
	try{
		const W F = load_word(pair), R = load_word(pair + 2);
		std::vector<Amplicon> amp;
		std::map<uint32_t, float> fi, ri;
		collect_candidates(amp, fi, ri, F, R, *s, s->opt.target_threshold*s->opt.search_multiplier,
			s->opt.amp_min, s->opt.amp_max);                                      // assay.h:405-408
		update_identity(fi, F, s->keys, s->opt.use_taq_mama != 0);
		update_identity(ri, R, s->keys, s->opt.use_taq_mama != 0);
		if(amp.empty()) return 0;                                                 // pcr_assay.cpp:271-302
		double ret = 0.0;
		std::unordered_set<uint32_t> valid;
		for(const Amplicon &a : amp){
			const float local = sqrtf(fi[a.f]*ri[a.r]);
			if(local >= s->opt.target_threshold && valid.find(a.index) == valid.end()){
				valid.insert(a.index);
				ret += a.weight;
			}
		}
		return (float)ret;
	}
	catch(const char *e){ s->err = e; return -1.0f; }
}

// One local-search move evaluated the way optimize_pcr.cpp does it (e.g. :61-100 for +degeneracy; the
// other five moves follow the same pattern): the candidate amplicons are those of the BASE pair,
// collected at target_threshold*search_multiplier (optimize.cpp:61-63, assay.h:405-408); for a variant
// of one oligo only that oligo's identity table is recomputed (update_identity with the variant word:
// its own length, its own 3' bases), then compute_coverage at target_threshold (pcr_assay.cpp:271-302).
// orient_out[v*T + i]: bit 0 = sequence i amplified through an {F(+),R(-)} candidate, bit 1 = {R(+),F(-)}.
int orc_session_move_coverage(orc_session *s, const uint64_t base[4], int side, const uint64_t *variants, unsigned n_variants,
	float *cov_out, unsigned char *orient_out)
{
	try{
		const W F = load_word(base), R = load_word(base + 2);
		const size_t T = s->seq.size();
		std::vector<Amplicon> amp;
		std::map<uint32_t, float> fi, ri;
		collect_candidates(amp, fi, ri, F, R, *s, s->opt.target_threshold*s->opt.search_multiplier, s->opt.amp_min, s->opt.amp_max);
		update_identity(fi, F, s->keys, s->opt.use_taq_mama != 0);               // update_target_candidates, assay.h:436-440
		update_identity(ri, R, s->keys, s->opt.use_taq_mama != 0);
		if(orient_out) memset(orient_out, 0, T*n_variants);
		for(unsigned v = 0;v < n_variants;++v){
			const W var = load_word(variants + 2*v);
			update_identity(side == 0 ? fi : ri, var, s->keys, s->opt.use_taq_mama != 0);
			double ret = 0.0;
			std::unordered_set<uint32_t> valid;
			for(const Amplicon &a : amp){
				const float local = sqrtf(fi[a.f]*ri[a.r]);
				if(local >= s->opt.target_threshold){
					if(orient_out) orient_out[(size_t)v*T + a.index] |= a.orient;
					if(valid.find(a.index) == valid.end()){ valid.insert(a.index); ret += a.weight; }
				}
			}
			cov_out[v] = amp.empty() ? 0.0f : (float)ret;
		}
		return 0;
	}
	catch(const char *e){ s->err = e; return -1; }
}

extern "C" int orc_is_valid(const uint64_t word[2], float salt, float primer_strand, float tm_min, float tm_max,
	float max_hairpin, float max_dimer, int check_homo_dimer);

namespace {
// compute_coverage (pcr_assay.cpp:271-302)
float coverage_of(const std::vector<Amplicon> &amp, std::map<uint32_t, float> &fi, std::map<uint32_t, float> &ri, float thr)
{
	if(amp.empty()) return 0;
	double ret = 0.0;
	std::unordered_set<uint32_t> valid;
	for(const Amplicon &a : amp){
		const float local = sqrtf(fi[a.f]*ri[a.r]);
		if(local >= thr && valid.find(a.index) == valid.end()){ valid.insert(a.index); ret += a.weight; }
	}
	return (float)ret;
}
struct ScoreO {                                                                   // pcramp.h:158-208
	float tc = -1.0e6f, bc = 1.0e6f, ov = 0.0f;
	float accuracy() const { return tc - bc; }
	bool gt(const ScoreO &r) const { return (accuracy() == r.accuracy()) ? (ov > r.ov) : (accuracy() > r.accuracy()); }
	bool lt(const ScoreO &r) const { return (accuracy() == r.accuracy()) ? (ov < r.ov) : (accuracy() < r.accuracy()); }
	bool eq(const ScoreO &r) const { return accuracy() == r.accuracy() && ov == r.ov; }
};
}

namespace {
// Candidates, identity tables and score of the base assay (optimize.cpp:61-79)
struct MoveState {
	std::vector<Amplicon> tamp, bamp;
	std::map<uint32_t, float> tfi, tri, bfi, bri, mfi, mri;
	ScoreO base;
};

float max_overlap(const W &a, const W &b);                                        // defined below (oligo reuse)

// use_multiplex: the keys of the multiplex background DB (main.cpp:989-1001) and the assays designed so far
struct Mx { std::vector<W> keys; std::vector<W> pool; /* F0, R0, F1, R1, ... */ };

const float REUSE_BONUS = 10.0f;                                                   // MULTIPLEX_OLIGO_REUSE_BONUS, assay.h:19

// compute_multiplex_background_coverage, pcr_assay.cpp:304-336
float multiplex_coverage_of(const std::map<uint32_t, float> &fi, const std::map<uint32_t, float> &ri, float thr)
{
	if(fi.empty() && ri.empty()) return 0;
	double ret = 0.0;
	std::set<uint32_t> valid;
	for(const auto &kv : fi){ if(kv.second >= thr && !valid.count(kv.first)){ valid.insert(kv.first); ret += 1.0; } }
	for(const auto &kv : ri){ if(kv.second >= thr && !valid.count(kv.first)){ valid.insert(kv.first); ret += 1.0; } }
	return (float)ret;
}

float pool_overlap(const W &w, const std::vector<W> &pool, float start)
{
	float r = start;
	for(const W &p : pool) r = std::max(r, max_overlap(w, p));                    // F then R of every pooled assay
	return r;
}

void move_state(MoveState &m, orc_session *t, orc_session *b, const W &F, const W &R, const orc_move_options *mo, const Mx *mx = nullptr)
{
	const bool taq = t->opt.use_taq_mama != 0;
	collect_candidates(m.tamp, m.tfi, m.tri, F, R, *t, t->opt.target_threshold*t->opt.search_multiplier, t->opt.amp_min, t->opt.amp_max);
	m.bamp.clear(); m.bfi.clear(); m.bri.clear();
	if(b && !b->keys.empty())                                                         // assay.h:411-421
		collect_candidates(m.bamp, m.bfi, m.bri, F, R, *b, mo->bg_threshold*mo->bg_multiplier, mo->bg_amp_min, mo->bg_amp_max);
	update_identity(m.tfi, F, t->keys, taq); update_identity(m.tri, R, t->keys, taq);
	if(b){ update_identity(m.bfi, F, b->keys, taq); update_identity(m.bri, R, b->keys, taq); }
	m.base.tc = coverage_of(m.tamp, m.tfi, m.tri, t->opt.target_threshold);
	m.base.bc = coverage_of(m.bamp, m.bfi, m.bri, mo->bg_threshold);
	m.base.ov = 0.0f;
	m.mfi.clear(); m.mri.clear();
	if(mx){                                                                           // optimize.cpp:79-97
		if(!mx->keys.empty()){                                                        // collect_multiplex_background_candidates, pcr_assay.cpp:71-102
			std::vector<uint32_t> k;
			match_words(k, F, mx->keys, mo->bg_threshold);
			for(uint32_t i : k) m.mfi[i] = 0.0f;
			k.clear();
			match_words(k, R, mx->keys, mo->bg_threshold);
			for(uint32_t i : k) m.mri[i] = 0.0f;
		}
		update_identity(m.mfi, F, mx->keys, taq); update_identity(m.mri, R, mx->keys, taq);
		m.base.bc += multiplex_coverage_of(m.mfi, m.mri, mo->bg_threshold);
		const float bf = pool_overlap(F, mx->pool, 0.0f), br = pool_overlap(R, mx->pool, 0.0f);   // compute_oligo_overlap, pcr_assay.cpp:736-754
		m.base.ov = ((bf == 1.0f) ? REUSE_BONUS : bf) + ((br == 1.0f) ? REUSE_BONUS : br);
	}
}

// One move function of optimize_pcr.cpp for oligo `side` of (F, R); `thr` = m_score_threshold
void move_eval(MoveState &m, orc_session *t, orc_session *b, const W &F, const W &R, int move, int side,
	const orc_move_options *mo, const ScoreO &thr, W &best_w, ScoreO &best, const Mx *mx = nullptr)
{
	const bool taq = t->opt.use_taq_mama != 0;
	const W cur = side == 0 ? F : R;
	// the trial words, in the reference's order, after its cheap gates (before is_valid)
	std::vector<W> trials;
	const int len = (int)cur.size();
	switch(move){
		case 0:                                                                       // increase_degeneracy, optimize_pcr.cpp:17-19,54-76
			if(cur.degeneracy() >= mo->degen) break;
			for(int i = cur.start();i <= cur.stop();++i){
				for(unsigned bb = 1;bb <= 8;bb <<= 1){
					if(cur.get(i) & bb) continue;
					W w = cur; w.set(cur.get(i) | bb, i);
					if(w.degeneracy() > mo->degen) continue;
					trials.push_back(w);
				}
			}
			break;
		case 1:                                                                       // decrease_degeneracy, :232-247
			for(int i = cur.start();i <= cur.stop();++i){
				const unsigned c = cur.get(i);
				for(unsigned bb = 1;bb <= 8;bb <<= 1){
					const unsigned d = c & ~bb;
					if(!d || d == c) continue;
					W w = cur; w.set(d, i); trials.push_back(w);
				}
			}
			break;
		case 2: if(len != mo->primer_min){ W w = cur; if(w.start() < 32) w.set(0, w.start()); trials.push_back(w); } break;   // trim5 :391-399, word.h:355
		case 3: if(len != mo->primer_min){ W w = cur; if(w.stop() >= 0) w.set(0, w.stop()); trials.push_back(w); } break;    // trim3, word.h:364
		case 4:                                                                       // grow5 :671-673,709-713, word.h:374
			if(len == mo->primer_max) break;
			for(unsigned bb = 1;bb <= 8;bb <<= 1){ W w = cur; const int i = w.start() - 1; if(i >= 0) w.set(bb, i); trials.push_back(w); }
			break;
		case 5:                                                                       // grow3, word.h:383
			if(len == mo->primer_max) break;
			for(unsigned bb = 1;bb <= 8;bb <<= 1){ W w = cur; const int i = w.stop() + 1; if(i < 32) w.set(bb, i); trials.push_back(w); }
			break;
		default: throw "unknown move";
	}
	best = ScoreO(); best_w.b[0] = best_w.b[1] = 0;
	// multiplex: the reuse term of the oligo that is NOT edited (e.g. optimize_pcr.cpp:27-53)
	float partial = 0.0f;
	if(mx){
		partial = pool_overlap(side == 0 ? R : F, mx->pool, 0.0f);
		if(partial == 1.0f) partial = REUSE_BONUS;
	}
	float carried_ov = 0.0f;   // increase_degeneracy never resets trial_score.oligo_overlap between trials (:133-145); the other moves do
	for(const W &w : trials){
		const int ok = orc_is_valid(w.b, mo->salt, mo->primer_strand, mo->tm_min, mo->tm_max, mo->max_hairpin, 0.0f, 0);
		if(ok < 0) throw "is_valid failed";
		if(!ok) continue;
		ScoreO tr;
		update_identity(side == 0 ? m.tfi : m.tri, w, t->keys, taq);
		tr.tc = coverage_of(m.tamp, m.tfi, m.tri, t->opt.target_threshold);
		const float bound = tr.tc + thr.bc - thr.tc;                                 // :95-97
		if(mx ? (bound < 0.0f) : (bound <= 0.0f)) continue;                           // :101-109
		if(b) update_identity(side == 0 ? m.bfi : m.bri, w, b->keys, taq);
		tr.bc = coverage_of(m.bamp, m.bfi, m.bri, mo->bg_threshold);
		if(mx){                                                                       // :111-145
			update_identity(side == 0 ? m.mfi : m.mri, w, mx->keys, taq);
			tr.bc += multiplex_coverage_of(m.mfi, m.mri, mo->bg_threshold);
			float ov = pool_overlap(w, mx->pool, (move == 0) ? carried_ov : 0.0f);
			ov = ((ov == 1.0f) ? REUSE_BONUS : ov) + partial;
			tr.ov = ov;
			carried_ov = ov;
		}
		if(tr.gt(best)){ best = tr; best_w = w; }
	}
	if(mx) update_identity(side == 0 ? m.mfi : m.mri, cur, mx->keys, taq);
	// the move functions restore the identity tables of the unmodified oligo before returning
	update_identity(side == 0 ? m.tfi : m.tri, cur, t->keys, taq);
	if(b) update_identity(side == 0 ? m.bfi : m.bri, cur, b->keys, taq);
}
}

int orc_optimization_move(orc_session *t, orc_session *b, const uint64_t pair[4], int move, int side,
	const orc_move_options *mo, uint64_t out_word[2], float out_score[3], float base_score_out[2])
{
	try{
		const W F = load_word(pair), R = load_word(pair + 2);
		MoveState m;
		move_state(m, t, b, F, R, mo);
		if(base_score_out){ base_score_out[0] = m.base.tc; base_score_out[1] = m.base.bc; }
		W best_w; ScoreO best;
		move_eval(m, t, b, F, R, move, side, mo, m.base, best_w, best);
		out_word[0] = best_w.b[0]; out_word[1] = best_w.b[1];
		out_score[0] = best.tc; out_score[1] = best.bc; out_score[2] = best.ov;
		return 0;
	}
	catch(const char *e){ t->err = e; return -1; }
}

// optimize() (optimize.cpp:14-207), non-multiplex: greedy local search over `moves` for both oligos.
// pair_inout receives the best assay; returns its score.
static int optimize_impl(orc_session *t, orc_session *b, const Mx *mx, uint64_t pair_inout[4], const int *moves, int n_moves,
	const orc_move_options *mo, float out_score[3], int *iterations_out)
{
	try{
		W bestF = load_word(pair_inout), bestR = load_word(pair_inout + 2);
		W aF = bestF, aR = bestR;
		ScoreO best_score, approx_score;
		std::set<std::pair<W, W> > previous;
		previous.insert(std::make_pair(bestF, bestR));
		int iteration = 0;
		while(true){
			bool improved = false;
			++iteration;
			MoveState m;
			move_state(m, t, b, aF, aR, mo, mx);                                      // :61-97
			approx_score = m.base;
			if(approx_score.lt(best_score)) break;                                    // :99-105
			best_score = approx_score; bestF = aF; bestR = aR;
			W local_seq; local_seq.b[0] = local_seq.b[1] = 0;
			int local_oligo = -1;
			ScoreO local_score = approx_score;
			for(int side = 0;side < 2;++side){                                        // :120-141
				for(int k = 0;k < n_moves;++k){
					W w; ScoreO sc;
					move_eval(m, t, b, aF, aR, moves[k], side, mo, local_score, w, sc, mx);
					if(sc.gt(local_score) || (sc.eq(local_score) && w.degeneracy() < local_seq.degeneracy())){
						local_score = sc; local_seq = w; local_oligo = side; improved = true;
					}
				}
			}
			if(!improved) break;
			approx_score = local_score;
			local_seq.center();                                                       // :152
			if(local_oligo == 0) aF = local_seq; else aR = local_seq;
			if(previous.find(std::make_pair(aF, aR)) != previous.end()) break;        // :196-202
			previous.insert(std::make_pair(aF, aR));
		}
		pair_inout[0] = bestF.b[0]; pair_inout[1] = bestF.b[1]; pair_inout[2] = bestR.b[0]; pair_inout[3] = bestR.b[1];
		out_score[0] = best_score.tc; out_score[1] = best_score.bc; out_score[2] = best_score.ov;
		if(iterations_out) *iterations_out = iteration;
		return 0;
	}
	catch(const char *e){ t->err = e; return -1; }
}

extern "C" float orc_max_dimer_tm(const uint64_t pair[4], float salt, float primer_strand);

// make_degenerate (optimize.cpp:356-398) -> PCR::maximize_degeneracy (pcr_assay.cpp:111-230): the top-down start of the
// local search.  Candidates and identities of the assay (collect_target_candidates + update_target_candidates), the
// amplicons sorted by sqrtf(f identity x r identity) descending (std::sort with assay.h:166-195's comparator on the
// reference's amplicon order: the same libstdc++ algorithm, so the same permutation of ties), then the two greedy phases.
int orc_make_degenerate(orc_session *t, uint64_t pair_inout[4], const orc_move_options *mo, float max_dimer, int *valid_out)
{
	try{
		W F = load_word(pair_inout), R = load_word(pair_inout + 2);
		std::vector<Amplicon> amp;
		std::map<uint32_t, float> fi, ri;
		collect_candidates(amp, fi, ri, F, R, *t, t->opt.target_threshold*t->opt.search_multiplier, t->opt.amp_min, t->opt.amp_max);   // assay.h:401-408
		update_identity(fi, F, t->keys, t->opt.use_taq_mama != 0);
		update_identity(ri, R, t->keys, t->opt.use_taq_mama != 0);
		std::sort(amp.begin(), amp.end(), [&](const Amplicon &l, const Amplicon &r){
			return sqrtf(fi[l.f]*ri[l.r]) > sqrtf(fi[r.f]*ri[r.r]); });                   // assay.h:176-194
		auto unite = [](const W &w, const W &key){                                    // Word::operator|, word.h:421-440
			W r = w;
			const int first = w.start(), last = w.stop();
			for(int i = first;i <= last;++i){ const unsigned b = key.get(i); if(b) r.set(r.get(i) | b, i); }
			return r;
		};
		auto valid = [&](const W &w){
			return orc_is_valid(w.b, mo->salt, mo->primer_strand, mo->tm_min, mo->tm_max, mo->max_hairpin, max_dimer, 1) == 1;
		};
		for(const Amplicon &a : amp){                                                 // pcr_assay.cpp:113-130
			const W lf = unite(F, t->keys[a.f]), lr = unite(R, t->keys[a.r]);
			if(lf.degeneracy() <= (double)mo->degen && valid(lf)) F = lf;
			if(lr.degeneracy() <= (double)mo->degen && valid(lr)) R = lr;
		}
		auto dimer = [&](const W &f, const W &r){ const uint64_t p[4] = {f.b[0], f.b[1], r.b[0], r.b[1]}; return orc_max_dimer_tm(p, mo->salt, mo->primer_strand); };
		float min_tm = dimer(F, R);                                                   // :132-229
		const int last_f = F.stop(), last_r = R.stop();
		int ok = 1;
		while(min_tm > max_dimer){
			float curr = 1.0e6f; int best_o = -1; W best;
			for(int side = 0;side < 2;++side){
				W &w = side ? R : F;
				for(int i = w.start();i <= (side ? last_r : last_f);++i){
					const unsigned cur = w.get(i);
					for(unsigned b = 1;b <= 8;b <<= 1){
						const unsigned d = cur & ~b;
						if(!d || d == cur) continue;
						w.set(d, i);
						const float tm = dimer(F, R);
						if(tm < curr){ curr = tm; best_o = side; best = w; }
						w.set(cur, i);
					}
				}
			}
			if(best_o < 0){ ok = 0; break; }
			if(best_o == 0) F = best; else R = best;
			min_tm = curr;
		}
		pair_inout[0] = F.b[0]; pair_inout[1] = F.b[1]; pair_inout[2] = R.b[0]; pair_inout[3] = R.b[1];
		if(valid_out) *valid_out = ok;
		return 0;
	}
	catch(const char *e){ t->err = e; return -1; }
}

int orc_optimize(orc_session *t, orc_session *b, uint64_t pair_inout[4], const int *moves, int n_moves,
	const orc_move_options *mo, float out_score[3], int *iterations_out)
{
	return optimize_impl(t, b, nullptr, pair_inout, moves, n_moves, mo, out_score, iterations_out);
}

static void build_mx(Mx &mx, orc_session *amplicons, const uint64_t *pool, unsigned n_pool)
{
	std::vector<Entry> db;
	for(size_t i = 0;i < amplicons->seq.size();++i)
		pack(amplicons->seq[i], (unsigned)i, amplicons->opt.pack_max_degen, 0.0f, 1.0f, (unsigned)amplicons->opt.min_primer, db);
	std::stable_sort(db.begin(), db.end(), entry_key_less);
	for(const Entry &e : db){ if(mx.keys.empty() || !(mx.keys.back() == e.w)) mx.keys.push_back(e.w); }
	for(unsigned i = 0;i < n_pool;++i){ mx.pool.push_back(load_word(pool + 4*i)); mx.pool.push_back(load_word(pool + 4*i + 2)); }
}

// One move with opt.use_multiplex, the base Score (incl. the multiplex terms) as the score threshold.
int orc_optimization_move_multiplex(orc_session *t, orc_session *b, orc_session *amplicons, const uint64_t *pool, unsigned n_pool,
	const uint64_t pair[4], int move, int side, const orc_move_options *mo, uint64_t out_word[2], float out_score[3], float base_score_out[3])
{
	try{
		Mx mx;
		build_mx(mx, amplicons, pool, n_pool);
		const W F = load_word(pair), R = load_word(pair + 2);
		MoveState m;
		move_state(m, t, b, F, R, mo, &mx);
		if(base_score_out){ base_score_out[0] = m.base.tc; base_score_out[1] = m.base.bc; base_score_out[2] = m.base.ov; }
		W best_w; ScoreO best;
		move_eval(m, t, b, F, R, move, side, mo, m.base, best_w, best, &mx);
		out_word[0] = best_w.b[0]; out_word[1] = best_w.b[1];
		out_score[0] = best.tc; out_score[1] = best.bc; out_score[2] = best.ov;
		return 0;
	}
	catch(const char *e){ t->err = e; return -1; }
}

// optimize() with opt.use_multiplex: `amplicons` = the accepted assays' amplicons (their pack is the multiplex
// background DB, main.cpp:989-1001), pool = n_pool x {F[2], R[2]} = the assays designed so far.
int orc_optimize_multiplex(orc_session *t, orc_session *b, orc_session *amplicons, const uint64_t *pool, unsigned n_pool,
	uint64_t pair_inout[4], const int *moves, int n_moves, const orc_move_options *mo, float out_score[3])
{
	Mx mx;
	build_mx(mx, amplicons, pool, n_pool);
	return optimize_impl(t, b, &mx, pair_inout, moves, n_moves, mo, out_score, nullptr);
}

float orc_weighted_coverage(orc_session *s, const unsigned char *bits)            // main.cpp:1402-1418
{
	double ret = 0.0;
	for(size_t i = 0;i < s->seq.size();++i){ if(bits[i]) ret += s->seq[i].weight; }
	return (float)ret;
}

} // extern "C"

// ------------------------------------------------------------------------------------ Smith-Waterman
namespace {

struct SWCell { int M, Iq, It, Ms_i, Ms_j, Iqs_i, Iqs_j, Its_i, Its_j; };

// One lane of SeqOverlap::align_smith_waterman (seq_overlap.cpp:347-609): match +2 / mismatch -3
// on IUPAC intersection, gap open -5, extend -2 (:60-67); start coordinates travel with the
// scores; the running maximum is taken with !(X.M < max), so ties go to the last cell in
// row-major order (:583).
void sw_align(const uint8_t *q, int qlen, const uint8_t *t, int tlen, orc_sw_result *out)
{
	const int MATCH = 2, MISMATCH = -3, GAP_OPEN = -5, GAP_EXT = -2;
	std::vector<SWCell> last(tlen + 1), curr(tlen + 1);
	for(int j = 0;j <= tlen;++j){                                                 // :383-392
		SWCell c = {0, GAP_OPEN, GAP_OPEN, 0, j, 0, 0, 0, 0};
		last[j] = c;
		SWCell z = {0, 0, 0, 0, 0, 0, 0, 0, 0};
		curr[j] = z;                                                              // :396 memset
	}
	int maxM = 0, max_si = 0, max_sj = 0, stop_i = 0, stop_j = 0;
	bool touched = false;
	for(int i = 0;i < qlen;++i){
		curr[0].M = 0; curr[0].Iq = curr[0].It = GAP_OPEN;                        // :404-411
		curr[0].Ms_i = i + 1; curr[0].Ms_j = 0;
		for(int j = 0;j < tlen;++j){
			const SWCell &A = last[j], &B = last[j + 1], &C = curr[j];
			SWCell X;
			const int tmp_c = std::max(std::max(A.M, A.Iq), A.It);                // :436
			const int s = ((q[i] & t[j]) > 0) ? MATCH : MISMATCH;                 // :440-444
			X.M = std::max(tmp_c, 0) + s;                                         // :455-458
			bool m = (A.M < A.Iq) || (A.M < A.It);                                // :483-488
			X.Ms_i = m ? 0 : A.Ms_i; X.Ms_j = m ? 0 : A.Ms_j;
			m = !(A.Iq < A.It) && (A.Iq > A.M);                                   // :491-503
			if(m){ X.Ms_i = A.Iqs_i; X.Ms_j = A.Iqs_j; }
			m = (A.It > A.M) && (A.It > A.Iq);                                    // :506-518
			if(m){ X.Ms_i = A.Its_i; X.Ms_j = A.Its_j; }
			if(0 > tmp_c){ X.Ms_i = i; X.Ms_j = j; }                              // :522-532
			int tb = std::max(C.M, 0) + GAP_OPEN, tc = std::max(C.Iq, 0) + GAP_EXT;   // :536-553
			X.Iq = std::max(tb, tc);
			if(tb < tc){ X.Iqs_i = C.Iqs_i; X.Iqs_j = C.Iqs_j; } else{ X.Iqs_i = C.Ms_i; X.Iqs_j = C.Ms_j; }
			tb = std::max(B.M, 0) + GAP_OPEN; tc = std::max(B.It, 0) + GAP_EXT;       // :557-574
			X.It = std::max(tb, tc);
			if(tb < tc){ X.Its_i = B.Its_i; X.Its_j = B.Its_j; } else{ X.Its_i = B.Ms_i; X.Its_j = B.Ms_j; }
			if(!(X.M < maxM)){                                                    // :579-603 (i < qlen, j < tlen hold)
				maxM = X.M; max_si = X.Ms_i; max_sj = X.Ms_j; stop_i = i; stop_j = j; touched = true;
			}
			curr[j + 1] = X;
		}
		std::swap(last, curr);
	}
	out->score = (int16_t)maxM;
	out->valid = touched ? 1 : 0;
	out->q_start = (int16_t)max_si; out->q_stop = (int16_t)stop_i;
	out->t_start = (int16_t)max_sj; out->t_stop = (int16_t)stop_j;
	out->last1 = out->last2 = 15;                                                 // seq_overlap.h:1266-1286
	if(touched && stop_j >= 1 && stop_j < tlen){ out->last1 = t[stop_j - 1]; out->last2 = t[stop_j]; }
	out->pad = 0;
}

// pack_query_slots / pack_target_slots(Word): size() slots starting at start() (seq_overlap.h:828-857,1099-1125)
int word_to_codes(const W &w, uint8_t *codes)
{
	const int len = (int)w.size();
	int j = w.start();
	for(int i = 0;i < len;++i, ++j){ codes[i] = (uint8_t)w.get(j); }
	return len;
}

struct SWWordResult { orc_sw_result r; };

void sw_words(const W &q, const W &t, orc_sw_result *out)
{
	uint8_t qc[32], tc[32];
	const int ql = word_to_codes(q, qc), tl = word_to_codes(t, tc);
	sw_align(qc, ql, tc, tl, out);
}

} // namespace

extern "C" {

void orc_sw_align(const uint8_t *q, int qlen, const uint8_t *t, int tlen, orc_sw_result *out) { sw_align(q, qlen, t, tlen, out); }

void orc_sw_align_words(const uint64_t q[2], const uint64_t t[2], orc_sw_result *out) { sw_words(load_word(q), load_word(t), out); }

int orc_session_background_match(orc_session *s, const uint64_t pair[4], float bg_threshold, float bg_multiplier,
	int amp_min, int amp_max, int use_taq_mama, int emulate_index_bug, unsigned char *bits_out)
{
	try{
		const W F = load_word(pair), R = load_word(pair + 2);
		memset(bits_out, 0, s->seq.size());
		std::vector<Amplicon> amp;
		std::map<uint32_t, float> fi, ri;
		if(!s->keys.empty()){                                                     // assay.h:411-421
			collect_candidates(amp, fi, ri, F, R, *s, bg_threshold*bg_multiplier, amp_min, amp_max);
		}
		float f_norm = 2.0f*F.size(), r_norm = 2.0f*R.size();                     // background_match.cpp:20-29
		if(f_norm > 0.0f) f_norm = 1.0f/f_norm;
		if(r_norm > 0.0f) r_norm = 1.0f/r_norm;
		const W Fc = F.complement(), Rc = R.complement();
		unsigned fp1 = 0, fp2 = 0, fm1 = 0, fm2 = 0, rp1 = 0, rp2 = 0, rm1 = 0, rm2 = 0;
		if(use_taq_mama){                                                         // :36-42 (get_last_two, word.h:299)
			fp1 = F.get(F.stop() - 1); fp2 = F.get(F.stop());
			fm1 = Fc.get(Fc.stop() - 1); fm2 = Fc.get(Fc.stop());
			rp1 = R.get(R.stop() - 1); rp2 = R.get(R.stop());
			rm1 = Rc.get(Rc.stop() - 1); rm2 = Rc.get(Rc.stop());
		}
		for(size_t k = 0;k < amp.size();++k){                                     // :66-164, one amplicon per 4 lanes
			const Amplicon &a = amp[k];
			// background_match.cpp:122 tests `(i+1) >= num_seq` (not the amplicon count): the odd-indexed
			// amplicon i+1 is silently skipped once i+1 reaches the number of SEQUENCES.  Reproduced only
			// on request, to pin this restatement against the compiled reference.
			if(emulate_index_bug && (k & 1) && k >= s->seq.size()) continue;
			orc_sw_result l0, l1, l2, l3;
			sw_words(F, s->keys[a.f], &l0);                                       // slot 0: F   + f[i]
			sw_words(Fc, s->keys[a.f], &l1);                                      // slot 1: (F) + f[i]
			sw_words(R, s->keys[a.r], &l2);                                       // slot 2: R   + r[i]
			sw_words(Rc, s->keys[a.r], &l3);                                      // slot 3: (R) + r[i]
			float FpRm = l0.score*l3.score*f_norm*r_norm;                         // :82-83 (int product, then float)
			float RpFm = l1.score*l2.score*f_norm*r_norm;
			if(use_taq_mama){                                                     // :85-93
				FpRm *= taq_mama(fp1, fp2, l0.last1, l0.last2)*taq_mama(rm1, rm2, l3.last1, l3.last2);
				RpFm *= taq_mama(rp1, rp2, l2.last1, l2.last2)*taq_mama(fm1, fm2, l1.last1, l1.last2);
			}
			float score;
			if(FpRm > RpFm) score = sqrt(FpRm); else score = sqrt(RpFm);          // :100-113 (double sqrt, stored as float)
			if(score >= bg_threshold) bits_out[a.index] = 1;                      // :116-120
		}
		return 0;
	}
	catch(const char *e){ s->err = e; return -1; }
}

int orc_session_multiplex_match(orc_session *s, const uint64_t pair[4], float bg_threshold, int use_taq_mama,
	unsigned char *bits_out)
{
	const W F = load_word(pair), R = load_word(pair + 2);
	memset(bits_out, 0, s->seq.size());
	float f_norm = 2.0f*F.size(), r_norm = 2.0f*R.size();                         // background_match.cpp:179-188
	if(f_norm > 0.0f) f_norm = 1.0f/f_norm;
	if(r_norm > 0.0f) r_norm = 1.0f/r_norm;
	const W Fc = F.complement(), Rc = R.complement();
	const W *oligo[4] = {&F, &Fc, &R, &Rc};
	uint8_t qc[4][32]; int ql[4];
	for(int k = 0;k < 4;++k) ql[k] = word_to_codes(*oligo[k], qc[k]);
	for(size_t i = 0;i < s->seq.size();++i){                                      // :224-294
		const Seq &q = s->seq[i];
		std::vector<uint8_t> t(q.len);
		for(uint64_t p = 0;p < q.len;++p) t[p] = (uint8_t)q.at(p);
		bool hit = false;
		for(int k = 0;k < 4;++k){
			orc_sw_result r;
			sw_align(qc[k], ql[k], t.data(), (int)q.len, &r);
			float sc = r.score*((k < 2) ? f_norm : r_norm);                       // :244-248
			if(use_taq_mama){                                                     // :250-257
				const W &o = *oligo[k];
				sc *= taq_mama(o.get(o.stop() - 1), o.get(o.stop()), r.last1, r.last2);
			}
			if(sc >= bg_threshold) hit = true;                                    // :260-267
		}
		if(hit) bits_out[i] = 1;
	}
	return 0;
}

} // extern "C"

// ------------------------------------------------------------------------------------ sampler (row f-2)
extern "C" unsigned orc_rand_r(unsigned *seed)                                    // glibc 2.35 stdlib/rand_r.c
{
	unsigned next = *seed;
	int result;
	next *= 1103515245u; next += 12345u;
	result = (int)((next/65536u) % 2048u);
	next *= 1103515245u; next += 12345u;
	result <<= 10; result ^= (int)((next/65536u) % 1024u);
	next *= 1103515245u; next += 12345u;
	result <<= 10; result ^= (int)((next/65536u) % 1024u);
	*seed = next;
	return (unsigned)result;
}

namespace {

int random_location(int start, int stop, unsigned *seed)                           // sample.cpp:6-12
{
	if(stop == start) throw "random_location: empty interval (the reference divides by zero)";
	return start + (int)orc_rand_r(seed) % (stop - start);
}

W subword(const Seq &s, int64_t loc, int len)                                     // sequence.cpp:269-302
{
	if(loc < 0 || (uint64_t)(loc + len) > s.len) throw "Sequence::subword: word is out of bounds";
	W r;
	for(int i = 0;i < len;++i) r.push_back(s.at((uint64_t)(loc + i)));             // an EOS nibble adds nothing
	return r;
}

bool valid_oligo(const W &w, const orc_sampler_options *o)
{
	const int r = orc_is_valid(w.b, o->salt, o->primer_strand, o->tm_min, o->tm_max, o->max_hairpin, o->max_dimer, 1);
	if(r < 0) throw "is_valid failed";
	return r == 1;
}

void random_assay(const std::vector<Seq> &seqs, unsigned *seed, const orc_sampler_options *o, W &f, W &r)   // pcr_assay.cpp:580-734
{
	std::vector<size_t> idx;
	for(size_t i = 0;i < seqs.size();++i){ if(seqs[i].active) idx.push_back(i); }  // :593-599
	if(idx.empty()) throw "PCR::random_assay: No active sequences found";
	const int span = o->primer_max - o->primer_min + 1;
	for(unsigned seq_iter = 1;;++seq_iter){
		if(seq_iter > 100) throw "PCR::random_assay: Unable to generate a valid initial assay to test!";   // :611-613
		const Seq &t = seqs[idx[orc_rand_r(seed) % idx.size()]];                   // :618
		const int len = (int)t.len;
		if(len < o->amp_min) throw "PCR::random_assay: sequence length is too small!";
		for(unsigned assay_iter = 1;assay_iter <= 100;++assay_iter){               // :626-634
			const int f_len = o->primer_min + (int)orc_rand_r(seed) % span;         // :637-638
			const int r_len = o->primer_min + (int)orc_rand_r(seed) % span;
			if(f_len + r_len > len) continue;
			const int f_start = random_location(0, (len + 1) - o->amp_min, seed);   // :644
			f = subword(t, f_start, f_len);
			if((int)f.size() != f_len) continue;                                    // :649
			if(f.degeneracy() > o->max_degen) continue;                             // :655
			if(!valid_oligo(f, o)) continue;                                        // :662
			const int r_start = random_location(f_start + o->amp_min - r_len,       // :677-679
				std::min((len + 1) - r_len, (f_start + o->amp_max + 1) - r_len), seed);
			const int amp_len = r_start - f_start + r_len;
			if(amp_len > o->amp_max || amp_len < o->amp_min) continue;              // :683-686
			r = subword(t, r_start, r_len).complement();                            // :688
			if((int)r.size() != r_len) continue;
			if(r.degeneracy() > o->max_degen) continue;                             // :697
			if(has_split(t, f_start, amp_len)) continue;                            // :703
			if(!valid_oligo(r, o)) continue;                                        // :710
			uint64_t pr[4] = {f.b[0], f.b[1], r.b[0], r.b[1]};
			if(orc_max_dimer_tm(pr, o->salt, o->primer_strand) > o->max_dimer) continue;   // :715
			f.center(); r.center();                                                 // :719
			return;
		}
	}
}

} // namespace

extern "C" int orc_random_assays(orc_session *s, unsigned *seed, unsigned n_trials, const orc_sampler_options *o, uint64_t *pairs_out)
{
	try{
		W f, r;                                                                     // each trial is a fresh PCR (main.cpp:525)
		for(unsigned t = 0;t < n_trials;++t){
			f = W(); r = W();
			random_assay(s->seq, seed, o, f, r);
			pairs_out[4*t] = f.b[0]; pairs_out[4*t + 1] = f.b[1]; pairs_out[4*t + 2] = r.b[0]; pairs_out[4*t + 3] = r.b[1];
		}
		return 0;
	}
	catch(const char *e){ s->err = e; return -1; }
}

// ------------------------------------------------------------------------------------ oligo reuse (multiplex Score term)
namespace {

// Word::max_overlap, word.h:38-91: one DP row over the subject; a cell takes its diagonal predecessor and adds
// one where the two nibbles are EQUAL (a mismatch keeps the count): the best ungapped diagonal, as a fraction
// of the longer word.
float max_overlap(const W &a, const W &b)
{
	unsigned char dp[32];
	memset(dp, 0, sizeof(dp));
	unsigned char max_score = 0;
	const int qs = a.start(), qe = a.stop(), ss = b.start(), se = b.stop();
	for(int i = qs;i <= qe;++i){
		unsigned char last = 0;
		const unsigned q = a.get(i);
		for(int j = ss;j <= se;++j){
			const unsigned char cur = dp[j];
			dp[j] = last;
			if(b.get(j) == q){ ++dp[j]; if(max_score < dp[j]) max_score = dp[j]; }
			last = cur;
		}
	}
	return float(max_score)/std::max(a.size(), b.size());
}

} // namespace

extern "C" float orc_word_max_overlap(const uint64_t a[2], const uint64_t b[2])
{
	W x, y; x.b[0] = a[0]; x.b[1] = a[1]; y.b[0] = b[0]; y.b[1] = b[1];
	return max_overlap(x, y);
}

extern "C" float orc_oligo_overlap(const uint64_t assay[4], const uint64_t *pool, unsigned n_pool)   // pcr_assay.cpp:736-754
{
	W f, r; f.b[0] = assay[0]; f.b[1] = assay[1]; r.b[0] = assay[2]; r.b[1] = assay[3];
	float best_f = 0.0f, best_r = 0.0f;
	for(unsigned i = 0;i < n_pool;++i){
		W pf, pr; pf.b[0] = pool[4*i]; pf.b[1] = pool[4*i + 1]; pr.b[0] = pool[4*i + 2]; pr.b[1] = pool[4*i + 3];
		best_f = std::max(best_f, max_overlap(f, pf)); best_f = std::max(best_f, max_overlap(f, pr));
		best_r = std::max(best_r, max_overlap(r, pf)); best_r = std::max(best_r, max_overlap(r, pr));
	}
	return ((best_f == 1.0f) ? 10.0f : best_f) + ((best_r == 1.0f) ? 10.0f : best_r);   // MULTIPLEX_OLIGO_REUSE_BONUS, assay.h:19
}

// ------------------------------------------------------------------------------------ multiplex background coverage
extern "C" int orc_multiplex_coverage(orc_session *s, const uint64_t base[4], int side, const uint64_t *variants, unsigned n_variants,
	float background_threshold, int use_taq_mama, float *cov_out, unsigned *n_keys_out)
{
	std::vector<Entry> db;                                                        // main.cpp:989-998
	for(size_t i = 0;i < s->seq.size();++i) pack(s->seq[i], (unsigned)i, s->opt.pack_max_degen, 0.0f, 1.0f, (unsigned)s->opt.min_primer, db);
	std::stable_sort(db.begin(), db.end(), entry_key_less);
	std::vector<W> keys;                                                          // keys(), pcramp.h:231-256
	for(const Entry &e : db){ if(keys.empty() || !(keys.back() == e.w)) keys.push_back(e.w); }
	if(n_keys_out) *n_keys_out = (unsigned)keys.size();
	const W F = load_word(base), R = load_word(base + 2);
	std::map<uint32_t, float> fi, ri;
	if(!keys.empty()){                                                            // collect_multiplex_background_candidates, pcr_assay.cpp:71-102
		std::vector<uint32_t> m;
		match_words(m, F, keys, background_threshold);
		for(uint32_t k : m) fi[k] = 0.0f;
		m.clear();
		match_words(m, R, keys, background_threshold);
		for(uint32_t k : m) ri[k] = 0.0f;
	}
	update_identity(fi, F, keys, use_taq_mama != 0);                              // update_multiplex_background_candidates, assay.h:448-452
	update_identity(ri, R, keys, use_taq_mama != 0);
	for(unsigned v = 0;v < n_variants;++v){
		const W w = load_word(variants + 2*v);
		update_identity(side == 0 ? fi : ri, w, keys, use_taq_mama != 0);         // optimize_pcr.cpp:111-119
		double ret = 0.0;                                                         // compute_multiplex_background_coverage, pcr_assay.cpp:304-336
		std::set<uint32_t> valid;
		for(const auto &kv : fi){ if(kv.second >= background_threshold && !valid.count(kv.first)){ valid.insert(kv.first); ret += 1.0; } }
		for(const auto &kv : ri){ if(kv.second >= background_threshold && !valid.count(kv.first)){ valid.insert(kv.first); ret += 1.0; } }
		cov_out[v] = (float)ret;
	}
	return 0;
}

// ------------------------------------------------------------------------------------ amplicons of an accepted assay (row f-3)
namespace {

const char BASE_LETTER[17] = "-ACMGRSVTWYHKDBN";                                  // bits_to_base, base_table.h:78-123 (for the sort order of the amplicon strings)

// PCR::extract_amplicon_seq, pcr_assay.cpp:443-542
void extract_amplicon_seq(std::vector<std::string> &amps, std::vector<unsigned> *bounds, const std::vector<OligoMatch> &m,
	uint8_t plus_o, uint8_t minus_o, const W &plus_w, const W &minus_w, const orc_session &s, int amp_min, int amp_max)
{
	const int PAD = 4;                                                            // MULTIPLEX_AMPLICON_PADDING, pcramp.h:57
	const int ps = plus_w.start(), pe = plus_w.stop(), ms = minus_w.start(), me = minus_w.stop();
	for(size_t p = 0;p < m.size();++p){
		if(m[p].o != plus_o) continue;
		for(size_t q = p;q < m.size();++q){
			if(m[p].index != m[q].index) break;
			if(m[q].o != minus_o) continue;
			if(loc3(m[p], ps, pe) >= loc5(m[q], ms, me)) continue;                // :468-471
			const int amp_len = loc3(m[q], ms, me) - loc5(m[p], ps, pe) + 1;
			if(amp_len < amp_min) continue;
			if(amp_len > amp_max) break;
			const unsigned amp_start = (unsigned)(loc3(m[p], ps, pe) + 1 - PAD);  // :489-490 (unsigned there)
			const unsigned np_len = (unsigned)loc5(m[q], ms, me) - amp_start + 2*PAD;   // :492-494
			const Seq &t = s.seq[m[p].index];
			std::string a(np_len, '?');
			bool valid = true;
			for(unsigned i = 0;i < np_len;++i){
				const uint64_t pos = (uint64_t)amp_start + i;
				const unsigned b = (pos < t.len) ? t.at(pos) : (unsigned)EOS;         // past the end: undefined there; an EOS here
				if(b == EOS){ valid = false; break; }
				a[i] = BASE_LETTER[b];
			}
			if(!valid) break;                                                     // :517-521
			amps.push_back(a);
			if(bounds){ bounds->push_back(m[p].index); bounds->push_back((unsigned)loc5(m[p], ps, pe)); bounds->push_back((unsigned)loc3(m[q], ms, me)); }
		}
	}
}

} // namespace

extern "C" long orc_session_collect_amplicons(orc_session *s, const uint64_t pair[4], float threshold, int amp_min, int amp_max,
	unsigned *bounds_out, long cap_bounds, unsigned char *amp_codes_out, long cap_codes, unsigned *amp_len_out, long cap_amp,
	long *n_amp_out)
{
	const W F = load_word(pair), R = load_word(pair + 2);
	std::vector<uint32_t> fm, rm;
	match_words(fm, F, s->keys, threshold*threshold);                             // :775-776
	match_words(rm, R, s->keys, threshold*threshold);
	std::vector<std::string> amps;
	std::vector<unsigned> bounds;
	std::vector<OligoMatch> om;
	find_oligo_match(om, fm, 0, PLUS, *s);
	find_oligo_match(om, rm, 1, MINUS, *s);
	std::stable_sort(om.begin(), om.end(), om_less);
	extract_amplicon_seq(amps, &bounds, om, 0, 1, F, R, *s, amp_min, amp_max);
	om.clear();
	find_oligo_match(om, fm, 0, MINUS, *s);
	find_oligo_match(om, rm, 1, PLUS, *s);
	std::stable_sort(om.begin(), om.end(), om_less);
	extract_amplicon_seq(amps, &bounds, om, 1, 0, R, F, *s, amp_min, amp_max);
	std::sort(amps.begin(), amps.end());                                          // :805-806
	amps.erase(std::unique(amps.begin(), amps.end()), amps.end());
	long used = 0, na = 0;
	for(const std::string &a : amps){
		if(na < cap_amp) amp_len_out[na] = (unsigned)a.size();
		for(char c : a){
			if(used < cap_codes) amp_codes_out[used] = (unsigned char)(strchr(BASE_LETTER, c) - BASE_LETTER);
			++used;
		}
		++na;
	}
	if(n_amp_out) *n_amp_out = na;
	const long nb = (long)(bounds.size()/3);
	for(long i = 0;i < nb && i < cap_bounds;++i){ bounds_out[3*i] = bounds[3*i]; bounds_out[3*i + 1] = bounds[3*i + 1]; bounds_out[3*i + 2] = bounds[3*i + 2]; }
	return (used > cap_codes || na > cap_amp) ? -3 : nb;
}
