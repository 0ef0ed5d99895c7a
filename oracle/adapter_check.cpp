// TEST INFRASTRUCTURE ONLY: the reference-side adapter of INTEGRATION.md section 1, with real bodies, compiled against
// the reference's own headers (assay.h, sequence.h, bitset.h, pcramp.h where they lie under /root/reference) and
// against include/pcramp_hip.h -- `make -C oracle adapter`.  It proves that the binding a maintainer would add
// type-checks against both sides: Word <-> pcr_word128, PCR <-> pcr_pair, deque<Sequence> -> pcr_load_sequences,
// BitSet <- bitset words, Options -> the argument structs, errors re-thrown as `const char*` (main.cpp:1269).
// Nothing here is copied from the reference; it only CALLS its public interface.
#include <deque>
#include <vector>
#include <string>
#include <string.h>

#include "pcramp_hip.h"
#include "assay.h"          // PCR, Word, Sequence, BitSet, Options, Score (reference headers)

struct DeviceScreen {       // one per process (one GPU); not thread-safe, like one NucCruc per thread (main.cpp:533,702)
	pcr_ctx *ctx;

	explicit DeviceScreen(const Options &opt, int device = 0)
	{
		pcr_params p = { opt.pack_max_degen, opt.pack_min_gc, opt.pack_max_gc };
		ctx = pcr_create(device, NULL, &p);
		if(!ctx) throw pcr_last_error();                 // reference error convention: throw const char*
	}
	~DeviceScreen() { pcr_destroy(ctx); }

	static pcr_word128 W(const Word &w)                  // Word is two u64 in exactly the ABI layout (word.h:673-679)
	{
		pcr_word128 r; unsigned char b[16]; w.mpi_pack(b); memcpy(r.w, b, 16); return r;
	}
	static Word W(const pcr_word128 &w)
	{
		Word r; unsigned char b[16]; memcpy(b, w.w, 16); r.mpi_unpack(b); return r;
	}
	static pcr_pair P(const PCR &a) { pcr_pair r = { W(a.oligo(FORWARD)), W(a.oligo(REVERSE)) }; return r; }
	static void check(int rc) { if(rc != PCR_OK) throw pcr_last_error(); }

	// after parse_fasta (main.cpp:257-344): hand the deque<Sequence> over once.  seq_buffer is private, so the codes are
	// read through Sequence::operator[] and re-packed two per byte, high nibble first (sequence.h:223-228).
	void load(pcr_set which, const std::deque<Sequence> &seq)
	{
		std::vector<uint64_t> off(seq.size()), len(seq.size());
		std::vector<float> weight(seq.size());
		std::vector<uint8_t> packed;
		for(size_t s = 0;s < seq.size();++s){
			off[s] = packed.size(); len[s] = seq[s].length(); weight[s] = seq[s].weight();
			for(size_t i = 0;i < seq[s].length();i += 2){
				const unsigned hi = seq[s][(unsigned)i], lo = (i + 1 < seq[s].length()) ? seq[s][(unsigned)(i + 1)] : 0u;
				packed.push_back((uint8_t)((hi << 4) | lo));
			}
		}
		check(pcr_load_sequences(ctx, which, packed.data(), off.data(), len.data(), weight.data(), (uint32_t)seq.size()));
		set_active(which, seq);
	}
	void set_active(pcr_set which, const std::deque<Sequence> &seq)                  // main.cpp:1105-1120
	{
		std::vector<uint8_t> a(seq.size());
		for(size_t s = 0;s < seq.size();++s) a[s] = seq[s].active() ? 1 : 0;
		check(pcr_set_active(ctx, which, a.data()));
	}
	void split(pcr_set which, unsigned idx, unsigned pos) { check(pcr_split(ctx, which, idx, pos)); }   // main.cpp:1008-1017

	// main.cpp:644-691 / 579-615: the per-iteration index build for all trial assays
	uint64_t select_words(pcr_set which, const std::vector<PCR> &trial, const Options &opt)
	{
		std::vector<pcr_pair> p(trial.size());
		for(size_t i = 0;i < trial.size();++i) p[i] = P(trial[i]);
		uint64_t n = 0;
		const bool bg = (which == PCR_SET_BACKGROUND);
		const float thr = bg ? opt.background_threshold*opt.background_search_multiplier : opt.target_threshold*opt.target_search_multiplier;
		const unsigned min_len = bg ? (unsigned)(opt.min_oligo_length()*0.9) : opt.min_oligo_length();   // main.cpp:595
		check(pcr_select_words(ctx, which, p.data(), (uint32_t)p.size(), opt.optimize_5, opt.optimize_3, thr, min_len, &n));
		return n;
	}

	static void to_bitset(BitSet &m, const std::vector<uint64_t> &w)
	{
		for(size_t i = 0;i < m.size();++i) m[i] = (w[i >> 6] >> (i & 63)) & 1u;
	}

	// main.cpp:898: best_assay.find_target_match(best_target_match, ...)
	void find_target_match(BitSet &m, const PCR &a, const Options &opt)
	{
		const pcr_pair p = P(a);
		pcr_amplify_args args = { opt.target_threshold, opt.target_threshold, opt.target_amplicon_range.first,
			opt.target_amplicon_range.second, opt.use_taq_mama ? 1 : 0 };
		std::vector<uint64_t> bits(pcr_bitset_words(ctx, PCR_SET_TARGET));
		check(pcr_amplify(ctx, PCR_SET_TARGET, &p, 1, &args, bits.data(), NULL, NULL, NULL));
		to_bitset(m, bits);
	}

	// optimize.cpp:61-77: collect + update + compute_target_coverage of one assay
	float target_coverage(const PCR &a, const Options &opt)
	{
		const pcr_pair p = P(a);
		pcr_amplify_args args = { opt.target_threshold*opt.target_search_multiplier, opt.target_threshold,
			opt.target_amplicon_range.first, opt.target_amplicon_range.second, opt.use_taq_mama ? 1 : 0 };
		float cov = 0.0f;
		check(pcr_amplify(ctx, PCR_SET_TARGET, &p, 1, &args, NULL, NULL, NULL, &cov));
		return cov;
	}

	// main.cpp:824: find_background_match (reference-identical amplicon pairing, background_match.cpp:122)
	void find_background_match(BitSet &m, const PCR &a, const Options &opt)
	{
		const pcr_pair p = P(a);
		pcr_background_args args = { opt.background_threshold*opt.background_search_multiplier, opt.background_threshold,
			opt.background_amplicon_range.first, opt.background_amplicon_range.second, opt.use_taq_mama ? 1 : 0, 0 };
		std::vector<uint64_t> bits(pcr_bitset_words(ctx, PCR_SET_BACKGROUND));
		check(pcr_background_match(ctx, PCR_SET_BACKGROUND, &p, 1, &args, bits.data()));
		to_bitset(m, bits);
	}

	static pcr_thermo_args thermo_args(const Options &opt)
	{
		pcr_thermo_args t = { opt.salt, opt.primer_strand, opt.primer_tm_range.first, opt.primer_tm_range.second, opt.max_hairpin, opt.max_dimer };
		return t;
	}

	// valid_pcr.cpp:5-45 for a batch of oligos (every move variant of optimize_pcr.cpp)
	std::vector<bool> is_valid(const std::vector<Word> &oligos, bool check_homo_dimer, const Options &opt)
	{
		std::vector<pcr_word128> w(oligos.size());
		for(size_t i = 0;i < oligos.size();++i) w[i] = W(oligos[i]);
		std::vector<pcr_thermo_result> r(oligos.size());
		const pcr_thermo_args t = thermo_args(opt);
		check(pcr_thermo(ctx, w.data(), (uint32_t)w.size(), check_homo_dimer ? 1 : 0, &t, r.data()));
		std::vector<bool> ok(oligos.size());
		for(size_t i = 0;i < oligos.size();++i) ok[i] = r[i].valid != 0;
		return ok;
	}

	// main.cpp:538-550 at one thread: the trial assays of a design iteration
	void random_assays(std::vector<PCR> &trial, unsigned *global_seed, const Options &opt)
	{
		uint32_t local = pcr_host_rand_r(global_seed);
		pcr_sampler_args sa = { opt.primer_range.first, opt.primer_range.second, opt.target_amplicon_range.first,
			opt.target_amplicon_range.second, opt.degen };
		const pcr_thermo_args t = thermo_args(opt);
		std::vector<pcr_pair> p(trial.size());
		check(pcr_random_assays(ctx, PCR_SET_TARGET, &local, (uint32_t)trial.size(), &sa, &t, p.data(), NULL));
		for(size_t i = 0;i < trial.size();++i){ trial[i].oligo(FORWARD, W(p[i].f)); trial[i].oligo(REVERSE, W(p[i].r)); }
	}

	// main.cpp:697-735: optimize() of every trial assay of the design iteration, in lockstep on the device; trial[t] becomes
	// the optimised assay, the returned Scores are optimize()'s (before the detailed background screening of :737-863)
	std::vector<Score> optimize_trials(std::vector<PCR> &trial, const std::deque<PCR> &pool, const std::deque<int> &moves,
		bool have_background, const Options &opt)
	{
		pcr_optimize_args o;
		memset(&o, 0, sizeof(o));
		o.max_degen = opt.degen; o.primer_min = opt.primer_range.first; o.primer_max = opt.primer_range.second;
		o.thermo = thermo_args(opt);
		const pcr_amplify_args ta = { opt.target_threshold*opt.target_search_multiplier, opt.target_threshold,
			opt.target_amplicon_range.first, opt.target_amplicon_range.second, opt.use_taq_mama ? 1 : 0 };
		const pcr_amplify_args ba = { opt.background_threshold*opt.background_search_multiplier, opt.background_threshold,
			opt.background_amplicon_range.first, opt.background_amplicon_range.second, opt.use_taq_mama ? 1 : 0 };
		o.target = ta; o.background = ba;
		o.have_background = have_background ? 1 : 0;
		o.use_multiplex = opt.use_multiplex ? 1 : 0; o.multiplex_threshold = opt.background_threshold;
		o.n_moves = 0;
		for(std::deque<int>::const_iterator m = moves.begin();m != moves.end() && o.n_moves < 8;++m) o.moves[o.n_moves++] = *m;
		std::vector<pcr_pair> in(trial.size()), best(trial.size()), pp(pool.size());
		for(size_t i = 0;i < trial.size();++i) in[i] = P(trial[i]);
		for(size_t i = 0;i < pool.size();++i) pp[i] = P(pool[i]);
		std::vector<float> sc(3*trial.size() + 3);
		check(pcr_optimize_batch(ctx, in.data(), (uint32_t)in.size(), &o, pp.data(), (uint32_t)pp.size(), best.data(), sc.data(), NULL));
		std::vector<Score> out(trial.size());
		for(size_t i = 0;i < trial.size();++i){
			trial[i].oligo(FORWARD, W(best[i].f)); trial[i].oligo(REVERSE, W(best[i].r));
			out[i].target_coverage = sc[3*i]; out[i].background_coverage = sc[3*i + 1]; out[i].oligo_overlap = sc[3*i + 2];
		}
		return out;
	}

	// main.cpp:709-721 (--optimize.top-down): make_degenerate of every trial assay before optimize(); ok[t] = its return value
	// (the reference drops the trial when it is false)
	std::vector<bool> make_degenerate(std::vector<PCR> &trial, const Options &opt)
	{
		pcr_optimize_args o;
		memset(&o, 0, sizeof(o));
		o.max_degen = opt.degen; o.primer_min = opt.primer_range.first; o.primer_max = opt.primer_range.second;
		o.thermo = thermo_args(opt);
		const pcr_amplify_args ta = { opt.target_threshold*opt.target_search_multiplier, opt.target_threshold,
			opt.target_amplicon_range.first, opt.target_amplicon_range.second, opt.use_taq_mama ? 1 : 0 };
		o.target = ta;
		std::vector<pcr_pair> p(trial.size());
		for(size_t i = 0;i < trial.size();++i) p[i] = P(trial[i]);
		std::vector<uint8_t> ok(trial.size() + 1);
		check(pcr_make_degenerate(ctx, p.data(), (uint32_t)p.size(), &o, ok.data()));
		std::vector<bool> out(trial.size());
		for(size_t i = 0;i < trial.size();++i){ trial[i].oligo(FORWARD, W(p[i].f)); trial[i].oligo(REVERSE, W(p[i].r)); out[i] = ok[i] != 0; }
		return out;
	}

	// main.cpp:744-803 for every trial assay: compatible[t] (:748-752) and the two background terms (:767-771, :786-803); the
	// loop over t then only applies `best_score < s` and `s.background_coverage <= opt.max_background_cover`
	void multiplex_screen(const std::vector<PCR> &trial, const std::deque<PCR> &pool, const Options &opt,
		std::vector<bool> &compatible, std::vector<float> &multiplex_cover, std::vector<float> &pool_cover)
	{
		pcr_multiplex_screen_args a;
		memset(&a, 0, sizeof(a));
		a.thermo = thermo_args(opt);
		a.background_threshold = opt.background_threshold; a.use_taq_mama = opt.use_taq_mama ? 1 : 0;
		a.target_threshold = opt.target_threshold;
		a.amp_min = opt.target_amplicon_range.first; a.amp_max = opt.target_amplicon_range.second;
		std::vector<pcr_pair> t(trial.size()), pp(pool.size());
		for(size_t i = 0;i < trial.size();++i) t[i] = P(trial[i]);
		for(size_t i = 0;i < pool.size();++i) pp[i] = P(pool[i]);
		std::vector<uint8_t> ok(trial.size() + 1);
		multiplex_cover.assign(trial.size() + 1, 0.0f); pool_cover.assign(trial.size() + 1, 0.0f);
		check(pcr_multiplex_screen(ctx, t.data(), (uint32_t)t.size(), pp.data(), (uint32_t)pp.size(), &a, NULL, ok.data(),
			multiplex_cover.data(), pool_cover.data()));
		compatible.resize(trial.size());
		for(size_t i = 0;i < trial.size();++i) compatible[i] = ok[i] != 0;
		multiplex_cover.resize(trial.size()); pool_cover.resize(trial.size());
	}

	// main.cpp:950-1113 in text form through the ABI's writer (the deflines keep their '>')
	std::string assay_text(const PCR &a, const std::deque<PCR> &pool)
	{
		const pcr_pair p = P(a);
		std::vector<pcr_pair> q(pool.size());
		for(size_t i = 0;i < pool.size();++i) q[i] = P(pool[i]);
		const int64_t n = pcr_format_oligos(&p, q.data(), (uint32_t)q.size(), 0, 1, NULL, 0);
		if(n < 0) throw pcr_last_error();
		std::string s((size_t)n + 1, '\0');
		pcr_format_oligos(&p, q.data(), (uint32_t)q.size(), 0, 1, &s[0], (uint64_t)n + 1);
		s.resize((size_t)n);
		return s;
	}

	// main.cpp:131-163, 440-443, 471-1264 in one call: the design loop on the loaded sets and the bytes `fout` receives.  What stays in
	// main() is Options::load, parse_fasta, load() of both sets and writing the returned text to opt.output_filename.
	std::string design(const Options &opt, const std::deque<Sequence> &targets, const std::deque<Sequence> &backgrounds, int argc, char *argv[])
	{
		pcr_design_args a;
		memset(&a, 0, sizeof(a));
		a.num_assay = opt.num_assay; a.num_trial = opt.num_trial; a.seed = opt.seed; a.top_down_search = opt.top_down_search ? 1 : 0;
		a.optimize_5 = opt.optimize_5 ? 1 : 0; a.optimize_3 = opt.optimize_3 ? 1 : 0;
		a.target_threshold = opt.target_threshold; a.target_search_multiplier = opt.target_search_multiplier;
		a.background_threshold = opt.background_threshold; a.background_search_multiplier = opt.background_search_multiplier;
		a.min_target_cover = opt.min_target_cover; a.max_background_cover = opt.max_background_cover;
		a.target_amp_min = opt.target_amplicon_range.first; a.target_amp_max = opt.target_amplicon_range.second;
		a.background_amp_min = opt.background_amplicon_range.first; a.background_amp_max = opt.background_amplicon_range.second;
		a.primer_min = opt.primer_range.first; a.primer_max = opt.primer_range.second; a.max_degen = opt.degen;
		a.thermo = thermo_args(opt); a.use_taq_mama = opt.use_taq_mama ? 1 : 0; a.use_multiplex = opt.use_multiplex ? 1 : 0;
		std::vector<std::string> td, bd;
		std::vector<const char *> tp, bp;
		std::vector<uint64_t> tl, bl;
		for(std::deque<Sequence>::const_iterator i = targets.begin();i != targets.end();++i){ td.push_back(i->defline()); tl.push_back(i->length()); }
		for(std::deque<Sequence>::const_iterator i = backgrounds.begin();i != backgrounds.end();++i){ bd.push_back(i->defline()); bl.push_back(i->length()); }
		for(size_t i = 0;i < td.size();++i) tp.push_back(td[i].c_str());
		for(size_t i = 0;i < bd.size();++i) bp.push_back(bd[i].c_str());
		pcr_output o;
		memset(&o, 0, sizeof(o));
		o.json = (opt.output_format == Options::JSON_OUTPUT) ? 1 : 0; o.use_multiplex = opt.use_multiplex ? 1 : 0;
		o.n_target = (uint32_t)td.size(); o.n_background = (uint32_t)bd.size();
		o.target_deflines = tp.data(); o.background_deflines = bp.data(); o.target_lengths = tl.data(); o.background_lengths = bl.data();
		check(pcr_design(ctx, &a, &o, argc, (const char *const *)argv, NULL, 0, NULL));
		uint64_t n = 0;
		const char *text = pcr_design_output(ctx, &n);
		return std::string(text, (size_t)n);
	}
};

// one function that instantiates every member, so that the whole adapter is compiled and linked, not only parsed
extern "C" int adapter_check_touch(int run)
{
	if(!run) return 0;
	Options opt;
	DeviceScreen d(opt);
	std::deque<Sequence> seq;
	d.load(PCR_SET_TARGET, seq);
	d.split(PCR_SET_TARGET, 0, 0);
	std::vector<PCR> trial(2);
	unsigned seed = 1;
	d.random_assays(trial, &seed, opt);
	d.select_words(PCR_SET_TARGET, trial, opt);
	BitSet m(0, false);
	d.find_target_match(m, trial[0], opt);
	d.find_background_match(m, trial[0], opt);
	const float c = d.target_coverage(trial[0], opt);
	const std::vector<bool> ok = d.is_valid(std::vector<Word>(1, trial[0].oligo(FORWARD)), true, opt);
	std::deque<int> moves(1, PCR_MOVE_TRIM5);
	const std::vector<Score> sc = d.optimize_trials(trial, std::deque<PCR>(), moves, false, opt);
	std::vector<bool> comp; std::vector<float> mc, pc;
	d.multiplex_screen(trial, std::deque<PCR>(), opt, comp, mc, pc);
	if(d.make_degenerate(trial, opt).size() != trial.size()) return -1;
	if(sc.size() != comp.size()) return -1;
	char *no_args[1] = { NULL };
	const std::string file = d.design(opt, seq, std::deque<Sequence>(), 0, no_args);
	return (int)c + (int)ok.size() + (int)d.assay_text(trial[0], std::deque<PCR>()).size() + (int)file.size();
}
