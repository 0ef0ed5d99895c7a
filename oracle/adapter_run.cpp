// TEST INFRASTRUCTURE ONLY: the reference-side adapter of INTEGRATION.md (oracle/adapter_check.cpp: DeviceScreen) EXECUTED
// with real reference objects.  One process holds
//   * deque<Sequence> / vector<PCR> / Options built with the reference's own constructors and setters,
//   * the reference's functions for the design iteration's steps (Sequence::pack + select_words + MULTIMAP::sort + keys,
//     PCR::find_target_match, collect/update/compute_target_coverage, PCR::find_background_match, PCR::is_valid, make_degenerate(),
//     optimize())
//     -- main.cpp:579-691, 898, 824; optimize.cpp:14-207; valid_pcr.cpp:5 --
//   * and DeviceScreen forwarding the same objects to libpcramp_hip.so,
// and compares what comes back object for object: word-DB size, BitSet element by element, float coverage by value,
// is_valid flags, optimised Words and Scores.  So Word::mpi_pack's layout, the re-packing of Sequence through operator[],
// the bitset-word -> BitSet conversion and the Options -> argument-struct mapping are proven by running, not by compiling.
// Built into oracle/_ref/libadapter_check.so by `make -C oracle adapter` (needs /root/reference); the GPU box receives the
// prebuilt library.  Nothing is copied from the reference: this file only calls its public interface.
#include <cstdint>
#include <cstring>
#include <string>
#include <sstream>
#include <vector>
#include <deque>
#include <iostream>
#include <ostream>
#include <math.h>
#include <unordered_map>
#include <unordered_set>
#include <list>
#include <set>
#include <algorithm>

// PCR::background_amplicons is private; the harness reads its size only to skip the case where the reference itself reads
// stale SSE lanes (background_match.cpp:122), exactly as oracle/ref_harness.cpp does.
#define private public
#include "adapter_check.cpp"    // DeviceScreen (includes the reference's assay.h and include/pcramp_hip.h)
#undef private

namespace {

struct Rng {                                       // xorshift64*: the inputs are a function of the seed only
	uint64_t s;
	explicit Rng(uint64_t seed) : s(seed*0x9E3779B97F4A7C15ull + 0x1234567ull) {}
	uint64_t next() { s ^= s >> 12; s ^= s << 25; s ^= s >> 27; return s*0x2545F4914F6CDD1Dull; }
	unsigned below(unsigned n) { return (unsigned)((next() >> 33) % n); }
};

const char ACGT[5] = "ACGT";

std::string random_seq(Rng &r, unsigned len) { std::string s(len, 'A'); for(unsigned i = 0;i < len;++i) s[i] = ACGT[r.below(4)]; return s; }

std::string mutate(Rng &r, const std::string &root, unsigned per_mille)
{
	std::string s = root;
	for(size_t i = 0;i < s.size();++i){
		if(r.below(1000) < per_mille){ char c; do{ c = ACGT[r.below(4)]; } while(c == s[i]); s[i] = c; }
	}
	return s;
}

std::string revcomp(const std::string &s)
{
	std::string o(s.rbegin(), s.rend());
	for(size_t i = 0;i < o.size();++i){ switch(o[i]){ case 'A': o[i] = 'T'; break; case 'C': o[i] = 'G'; break; case 'G': o[i] = 'C'; break; case 'T': o[i] = 'A'; break; default: break; } }
	return o;
}

void make_sequences(std::deque<Sequence> &out, const std::vector<std::string> &txt)
{
	for(size_t i = 0;i < txt.size();++i){
		out.push_back(Sequence());
		Sequence &s = out.back();
		s = txt[i];                                 // Sequence::operator=(const string&), sequence.cpp
		s.weight(1.0f + 0.25f*(float)(i % 5));
		s.active(true);
	}
}

// main.cpp:644-691 (targets) / :579-615 (backgrounds) on the host, with the reference's own functions
size_t reference_db(const std::deque<Sequence> &seq, const std::vector<PCR> &trial, const Options &opt, bool background,
	MULTIMAP<Word, WordMatch> &db, std::vector<Word> &keys_out)
{
	db = MULTIMAP<Word, WordMatch>();
	const float thr = background ? opt.background_threshold*opt.background_search_multiplier : opt.target_threshold*opt.target_search_multiplier;
	const unsigned min_len = background ? (unsigned)(opt.min_oligo_length()*0.9) : (unsigned)opt.min_oligo_length();
	for(unsigned i = 0;i < seq.size();++i){
		if(!seq[i].active()) continue;
		MULTIMAP<Word, WordMatch> local_db;
		seq[i].pack(local_db, i, opt.pack_max_degen, opt.pack_min_gc, opt.pack_max_gc, min_len);
		select_words(db, local_db, trial, opt.optimize_5, opt.optimize_3, thr);
	}
	db.sort();
	keys_out = keys(db);
	return db.size();
}

void prefill(NucCruc &melt)                         // as oracle/ref_harness.cpp: the rings' stale element made defined
{
	for(unsigned i = 0;i < MAX_SEQUENCE_LENGTH;++i){ melt.query.push_back(BASE::E); melt.target.push_back(BASE::E); }
	melt.clear();
}

} // namespace

// stats[0] comparisons made, [1] amplification bits set (reference), [2] background bits set, [3] assays the search changed,
// [4] DB entries (targets), [5] DB entries (backgrounds), [6] is_valid verdicts compared, [7] of them true
// returns the number of MISMATCHES (0 = the adapter reproduces the reference), < 0 on an error (message on stderr)
extern "C" int adapter_run(unsigned seed, unsigned n_families, unsigned per_family, unsigned length, unsigned n_trials, long long *stats)
{
	long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	int bad = 0;
	try{
		Rng rng(seed);
		// ---- inputs: target families, backgrounds derived from the roots, trial assays cut from the targets (+ damaged copies)
		std::vector<std::string> ttxt, btxt;
		for(unsigned f = 0;f < n_families;++f){
			const std::string root = random_seq(rng, length);
			for(unsigned m = 0;m < per_family;++m) ttxt.push_back(mutate(rng, root, 40));
			btxt.push_back(mutate(rng, root, 130));
			btxt.push_back(mutate(rng, root, 60));
		}
		btxt.push_back(random_seq(rng, length));
		std::deque<Sequence> target_seq, background_seq;
		make_sequences(target_seq, ttxt);
		make_sequences(background_seq, btxt);

		Options opt;
		opt.target_threshold = 0.9f; opt.target_search_multiplier = 0.9f;
		opt.background_threshold = 0.45f; opt.background_search_multiplier = 1.6f;      // low final threshold so that background bits are set
		opt.target_amplicon_range = std::make_pair(80, 200);
		opt.background_amplicon_range = std::make_pair(0, 2000);
		opt.use_taq_mama = false; opt.use_multiplex = false;
		opt.pack_max_degen = 256; opt.pack_min_gc = 0.0f; opt.pack_max_gc = 1.0f;
		opt.primer_range = std::make_pair(18, 25);
		opt.optimize_5 = true; opt.optimize_3 = true;
		opt.degen = 4;
		opt.salt = 0.05f; opt.primer_strand = 9.0e-7f;
		opt.primer_tm_range = std::make_pair(45.0f, 75.0f);
		opt.max_hairpin = 45.0f; opt.max_dimer = 45.0f;
		opt.output_filter = Options::SILENT;

		std::vector<PCR> trial(n_trials);
		for(unsigned t = 0;t < n_trials;++t){
			const std::string &src = ttxt[rng.below((unsigned)ttxt.size())];
			const unsigned fl = 18 + rng.below(8), rl = 18 + rng.below(8), amp = 90 + rng.below(100);
			const unsigned fs = rng.below((unsigned)src.size() - amp);
			std::string f = src.substr(fs, fl), r = revcomp(src.substr(fs + amp - rl, rl));
			if(t % 3 == 2){ f = mutate(rng, f, 80); r = mutate(rng, r, 80); }       // damaged primers the search repairs
			Word wf, wr;
			wf = f; wr = r;                              // Word::operator=(const string&): centred, as the sampler stores them
			trial[t].oligo(FORWARD, wf); trial[t].oligo(REVERSE, wr);
		}

		DeviceScreen dev(opt);
		dev.load(PCR_SET_TARGET, target_seq);
		dev.load(PCR_SET_BACKGROUND, background_seq);

		for(int round = 0;round < 2;++round){
			if(round == 1){
				// main.cpp:1008-1017, 1105-1120: EOS splits inside a used target, a matched target deactivated
				target_seq[1].split_sequence(length/2); dev.split(PCR_SET_TARGET, 1, length/2);
				target_seq[1].split_sequence(length/3); dev.split(PCR_SET_TARGET, 1, length/3);
				target_seq[0].active(false);
				dev.set_active(PCR_SET_TARGET, target_seq);
			}
			// ---- the per-iteration DB build
			MULTIMAP<Word, WordMatch> tdb, bdb;
			std::vector<Word> tkeys, bkeys;
			const size_t n_ref = reference_db(target_seq, trial, opt, false, tdb, tkeys);
			const size_t n_bref = reference_db(background_seq, trial, opt, true, bdb, bkeys);
			const uint64_t n_dev = dev.select_words(PCR_SET_TARGET, trial, opt);
			const uint64_t n_bdev = dev.select_words(PCR_SET_BACKGROUND, trial, opt);
			++st[0]; if(n_dev != n_ref){ ++bad; std::cerr << "adapter_run: target DB " << n_dev << " != " << n_ref << "\n"; }
			++st[0]; if(n_bdev != n_bref){ ++bad; std::cerr << "adapter_run: background DB " << n_bdev << " != " << n_bref << "\n"; }
			st[4] = (long long)n_ref; st[5] = (long long)n_bref;

			for(unsigned t = 0;t < n_trials;++t){
				// main.cpp:898
				BitSet m_ref, m_dev((unsigned)target_seq.size(), false);
				PCR p = trial[t];
				p.find_target_match(m_ref, tkeys, tdb, target_seq, opt);
				dev.find_target_match(m_dev, trial[t], opt);
				for(size_t i = 0;i < target_seq.size();++i){
					const bool r = (i < m_ref.size()) && m_ref[i];
					++st[0]; st[1] += r;
					if(r != (bool)m_dev[i]){ ++bad; std::cerr << "adapter_run: amplification bit of trial " << t << ", target " << i << " differs\n"; }
				}
				// optimize.cpp:61-77
				PCR q = trial[t];
				q.collect_target_candidates(tkeys, tdb, target_seq, opt);
				q.update_target_candidates(tkeys, opt.use_taq_mama);
				const float c_ref = q.compute_target_coverage(opt.target_threshold);
				const float c_dev = dev.target_coverage(trial[t], opt);
				++st[0]; if(c_ref != c_dev){ ++bad; std::cerr << "adapter_run: coverage of trial " << t << ": " << c_dev << " != " << c_ref << "\n"; }
				// main.cpp:824 (skipped where the reference itself reads stale SSE lanes: odd amplicon count below the sequence count)
				PCR b = trial[t];
				b.collect_background_candidates(bkeys, bdb, background_seq, opt);
				const size_t n_amp = b.background_amplicons.size();
				if(!((n_amp & 1) && n_amp < background_seq.size())){
					BitSet b_ref((unsigned)background_seq.size(), false), b_dev((unsigned)background_seq.size(), false);
					b.find_background_match(b_ref, bkeys, bdb, background_seq, opt, std::cerr);
					dev.find_background_match(b_dev, trial[t], opt);
					for(size_t i = 0;i < background_seq.size();++i){
						++st[0]; st[2] += (bool)b_ref[i];
						if((bool)b_ref[i] != (bool)b_dev[i]){ ++bad; std::cerr << "adapter_run: background bit of trial " << t << ", sequence " << i << " differs\n"; }
					}
				}
			}
		}

		// ---- valid_pcr.cpp:5-45 for every trial oligo
		{
			std::vector<Word> oligos;
			for(unsigned t = 0;t < n_trials;++t){ oligos.push_back(trial[t].oligo(FORWARD)); oligos.push_back(trial[t].oligo(REVERSE)); }
			const std::vector<bool> ok = dev.is_valid(oligos, true, opt);
			for(size_t i = 0;i < oligos.size();++i){
				NucCruc melt; prefill(melt); melt.salt(opt.salt);
				PCR p;
				const bool r = p.is_valid(FORWARD, oligos[i], melt, opt, true);
				++st[0]; ++st[6]; st[7] += r;
				if(r != ok[i]){ ++bad; std::cerr << "adapter_run: is_valid of oligo " << i << " differs\n"; }
			}
		}

		// ---- main.cpp:697-735: optimize() of every trial assay (the DBs of the last round are those of the trial batch)
		{
			MULTIMAP<Word, WordMatch> tdb, bdb;
			std::vector<Word> tkeys, bkeys;
			Options oo = opt;
			oo.background_threshold = 0.8f; oo.background_search_multiplier = 0.9f;     // the reference's defaults for the search
			reference_db(target_seq, trial, oo, false, tdb, tkeys);
			reference_db(background_seq, trial, oo, true, bdb, bkeys);
			dev.select_words(PCR_SET_TARGET, trial, oo);
			dev.select_words(PCR_SET_BACKGROUND, trial, oo);
			std::deque<int> moves;                                                   // main.cpp:82-95
			moves.push_back(PCR_MOVE_INCREASE_DEGENERACY); moves.push_back(PCR_MOVE_DECREASE_DEGENERACY); moves.push_back(PCR_MOVE_TRIM5);
			moves.push_back(PCR_MOVE_GROW5); moves.push_back(PCR_MOVE_TRIM3); moves.push_back(PCR_MOVE_GROW3);
			const Move mv[6] = { IncreaseDegeneracy, DecreaseDegeneracy, Trim5, Trim3, Grow5, Grow3 };
			std::vector<Move> ml;
			for(std::deque<int>::const_iterator m = moves.begin();m != moves.end();++m) ml.push_back(mv[*m]);
			std::vector<PCR> dev_trial = trial;
			const std::deque<PCR> pool;
			{   // main.cpp:709-721: the top-down start, every trial assay at once against the reference's make_degenerate one by one
				std::vector<PCR> deg = trial;
				const std::vector<bool> ok = dev.make_degenerate(deg, oo);
				for(unsigned t = 0;t < n_trials;++t){
					PCR p = trial[t];
					NucCruc melt; prefill(melt); melt.salt(oo.salt);
					std::ostringstream sink;
					const bool r_ok = make_degenerate(p, tkeys, tdb, target_seq, melt, oo, sink);
					++st[0];
					if(r_ok != ok[t] || !(p.oligo(FORWARD) == deg[t].oligo(FORWARD)) || !(p.oligo(REVERSE) == deg[t].oligo(REVERSE))){
						++bad; std::cerr << "adapter_run: make_degenerate of assay " << t << " differs\n";
					}
				}
			}
			const std::vector<Score> dsc = dev.optimize_trials(dev_trial, pool, moves, true, oo);
			const std::vector<Word> no_keys; const MULTIMAP<Word, WordMatch> no_db; const std::deque<Sequence> no_seq;
			for(unsigned t = 0;t < n_trials;++t){
				PCR p = trial[t];
				std::ostringstream sink;
				const Score rs = optimize(p, ml, tkeys, tdb, target_seq, bkeys, bdb, background_seq, no_keys, no_db, no_seq, pool, oo, sink);
				st[3] += !(p.oligo(FORWARD) == trial[t].oligo(FORWARD)) || !(p.oligo(REVERSE) == trial[t].oligo(REVERSE));
				++st[0];
				if(!(p.oligo(FORWARD) == dev_trial[t].oligo(FORWARD)) || !(p.oligo(REVERSE) == dev_trial[t].oligo(REVERSE))){
					++bad; std::cerr << "adapter_run: optimised assay " << t << " differs\n";
				}
				++st[0];
				if(rs.target_coverage != dsc[t].target_coverage || rs.background_coverage != dsc[t].background_coverage || rs.oligo_overlap != dsc[t].oligo_overlap){
					++bad; std::cerr << "adapter_run: Score of assay " << t << " differs: (" << dsc[t].target_coverage << ", " << dsc[t].background_coverage
						<< ") != (" << rs.target_coverage << ", " << rs.background_coverage << ")\n";
				}
			}
		}
	}
	catch(const char *e){ std::cerr << "adapter_run: " << e << "\n"; return -1; }
	catch(...){ std::cerr << "adapter_run: unknown exception\n"; return -2; }
	if(stats){ for(int i = 0;i < 8;++i) stats[i] = st[i]; }
	return bad;
}
