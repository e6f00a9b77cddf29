// Variant block: which variants are near enough to share k-mers, which chains of
// neighbours a variant can be combined with, and the signature k-mers of every
// allele.  Host-side restatement of VB (var_block.hpp:61-219, 408-786): this is
// the irregular, allocation-heavy part of the path ("Irregular
// enumeration"); it emits flat descriptors for mg_lookup_cover / mg_genotype, and
// blocks that hold one short-allele variant bypass it (mg_call_isolated).
#pragma once
#include <climits>
#include <cmath>
#include <exception>
#include <set>
#include <thread>
#include <unordered_map>

#include "io.hpp"

namespace malva {

// std::string(s, pos, n): clips at the end, throws when pos > size (as the reference would)
inline std::string substr_clip(const std::string &s, long pos, long n)
{
    if (pos < 0 || (size_t)pos > s.size()) throw std::out_of_range("reference window outside the contig");
    if (n < 0) n = 0;
    return s.substr((size_t)pos, (size_t)n);
}

// signatures[variant][allele] = list of signatures, each a list of k-mers  (VK_GROUP, var_block.hpp:33)
using AlleleSignatures = std::map<int, std::vector<std::vector<std::string>>>;

class Block {
  public:
    std::vector<Variant> vars;
    int k;
    explicit Block(int k_) : k(k_) {}
    struct PanelIndex;
    bool empty() const { return vars.empty(); }
    void clear() { vars.clear(); }
    void add(Variant &&v) { vars.push_back(std::move(v)); }

    // var_block.hpp:408-412
    static bool overlapping(const Variant &a, const Variant &b) { return a.ref_pos <= b.ref_pos && b.ref_pos < a.ref_pos + a.ref_size; }
    // var_block.hpp:417-423: v1.pos + v1.ref_size - v1.min_size - 1 + extra + ceil((float)k / 2) >= v2.pos -- in the
    // reference's arithmetic: under `using namespace std` that ceil is the float overload, so the int sum is converted
    // to FLOAT, the addition rounds to float and v2.pos is compared as a float.  Exact below 2^24; beyond (most of a
    // human chromosome) positions are rounded to multiples of 2..16 and the answer differs from the exact one now and
    // then, in both directions.  Monotone in both arguments (what the walk's early stop relies on).
    static bool near_f32(int lhs_sum, int k, int rhs_pos) { return (float)lhs_sum + std::ceil((float)k / 2) >= (float)rhs_pos; }
    bool near(const Variant &a, const Variant &b, int extra = 0) const { return near_f32(a.ref_pos + a.ref_size - a.min_size - 1 + extra, k, b.ref_pos); }
    bool near_to_last(const Variant &v) const { return near(vars.back(), v); } // var_block.hpp:77-80

    // One variant, every allele shorter than k: the only chain is the variant itself and each allele
    // carried by the panel has exactly one signature k-mer -- the case mg_call_isolated fuses on the device.
    bool is_lone_short() const
    {
        if (vars.size() != 1) return false;
        if (vars[0].ref_size >= k) return false;
        for (const auto &a : vars[0].alts)
            if ((int)a.size() >= k) return false;
        return vars[0].n_alleles() <= 64;
    }

    // get_combs_on_the_right (step +1) / _left (step -1), var_block.hpp:436-525, 534-624.  The two are
    // mirror images; ordered(x, y) puts the pair in genome order as the reference's argument order does.
    // px (optional): lets the walk stop where nothing can join any more -- the reference walks to the end of the
    // block whatever happens (O(B^2) per block); with sorted positions the chains are the same
    std::vector<std::vector<int>> chains(int i, int step, const PanelIndex *px = nullptr) const
    {
        const Variant &mid = vars[(size_t)i];
        std::vector<std::vector<int>> out;
        std::vector<int> sums;
        auto ov = [&](const Variant &x, const Variant &y) { return step > 0 ? overlapping(x, y) : overlapping(y, x); };
        auto nr = [&](const Variant &x, const Variant &y, int extra) { return step > 0 ? near(x, y, extra) : near(y, x, extra); };
        bool halt = false;
        for (int j = i + step; j >= 0 && j < (int)vars.size() && !halt; j += step) {
            const Variant &cur = vars[(size_t)j];
            if (px && px->sorted) {
                int max_sum = 0;
                for (int s_ : sums) max_sum = std::max(max_sum, s_);
                if (step > 0 ? !near_f32(mid.ref_pos + mid.ref_size - mid.min_size - 1 + max_sum, k, cur.ref_pos)
                             : !near_f32(cur.ref_pos + px->max_gain - 1 + max_sum, k, mid.ref_pos))
                    break;
            }
            if (!cur.is_present) continue;
            if (ov(mid, cur)) continue;
            const int gain = cur.ref_size - cur.min_size;
            if (out.empty()) {
                if (nr(mid, cur, 0)) {
                    out.push_back({j});
                    sums.push_back(gain);
                }
                continue;
            }
            bool added = false;
            for (size_t c = 0; c < out.size(); ++c) {
                if (!ov(vars[(size_t)out[c].back()], cur)) {
                    added = true;
                    if (nr(mid, cur, sums[c])) {
                        out[c].push_back(j);
                        sums[c] += gain;
                    }
                }
            }
            if (!added) {
                std::vector<std::vector<int>> fresh;
                std::vector<int> fresh_sums;
                for (size_t c = 0; c < out.size(); ++c) {
                    std::vector<int> nc = out[c];
                    int ns = sums[c];
                    // drop members that overlap cur (the reference indexes back() of an emptied vector
                    // here: undefined behaviour, restated as "stop when empty")
                    while (!nc.empty() && ov(vars[(size_t)nc.back()], cur)) {
                        const Variant &m = vars[(size_t)nc.back()];
                        ns -= m.ref_size - m.min_size;
                        nc.pop_back();
                    }
                    nc.push_back(j);
                    if (nr(mid, cur, ns)) {
                        added = true;
                        fresh.push_back(std::move(nc));
                        fresh_sums.push_back(ns + gain);
                    }
                }
                out.insert(out.end(), fresh.begin(), fresh.end());
                sums.insert(sums.end(), fresh_sums.begin(), fresh_sums.end());
                if (!added) halt = true;
            }
        }
        return out;
    }

    // combine_combs, var_block.hpp:630-677
    static std::vector<std::vector<int>> combine(const std::vector<std::vector<int>> &left, const std::vector<std::vector<int>> &right, int i)
    {
        std::vector<std::vector<int>> full;
        if (left.empty() && right.empty()) return {{i}};
        if (left.empty()) {
            for (const auto &r : right) {
                std::vector<int> c{i};
                c.insert(c.end(), r.begin(), r.end());
                full.push_back(std::move(c));
            }
            return full;
        }
        for (const auto &l : left) {
            std::vector<int> base(l.rbegin(), l.rend());
            base.push_back(i);
            if (right.empty()) full.push_back(base);
            else
                for (const auto &r : right) {
                    std::vector<int> c = base;
                    c.insert(c.end(), r.begin(), r.end());
                    full.push_back(std::move(c));
                }
        }
        return full;
    }

    // build_alleles_combs + combine_haplotypes, var_block.hpp:709-786.  A set of allele picks along the
    // chain; picks are compared by allele TEXT in the reference, i.e. by first index with that text.
    // What allele_picks needs from the panel, gathered once per block.  A panel has tens of thousands of samples
    // (C1: 27,934) and nearly all of them carry the reference allele at any given variant: a sample that is
    // all-reference along a chain picks allele 0 everywhere whatever its phasing, so only the samples that are
    // non-reference at some member of the chain have to be walked.
    struct PanelIndex {
        std::vector<std::vector<int>> nonref;  // per variant: samples with a non-reference allele (either haplotype)
        std::vector<std::vector<int>> canon;   // per variant, per allele: first allele index with the same text
        std::vector<uint8_t> any_unphased;     // per variant: some sample is unphased there
        std::vector<uint8_t> regular;          // per variant: genotypes and phasing both hold n_samples entries
        size_t n_samples = 0;
        bool sorted = true;                    // positions never decrease along the block
        int max_gain = 0;                      // max ref_size - min_size over the block
    };
    PanelIndex panel_index() const
    {
        PanelIndex px;
        const size_t B = vars.size();
        px.nonref.resize(B);
        px.canon.resize(B);
        px.any_unphased.assign(B, 0);
        px.regular.assign(B, 0);
        for (const Variant &v : vars)
            if (v.is_present) px.n_samples = std::max(px.n_samples, v.genotypes.size());
        for (size_t j = 0; j < B; ++j) {
            if (j && vars[j].ref_pos < vars[j - 1].ref_pos) px.sorted = false;
            px.max_gain = std::max(px.max_gain, vars[j].ref_size - vars[j].min_size);
        }
        for (size_t j = 0; j < B; ++j) {
            const Variant &v = vars[j];
            if (!v.is_present) continue;
            px.regular[j] = v.genotypes.size() == px.n_samples && v.phasing.size() == px.n_samples;
            px.canon[j].resize((size_t)v.n_alleles());
            for (int a = 0; a < v.n_alleles(); ++a) px.canon[j][(size_t)a] = v.allele_index(v.allele(a));
            for (size_t g = 0; g < v.genotypes.size(); ++g) {
                if (v.genotypes[g].first != 0 || v.genotypes[g].second != 0) px.nonref[j].push_back((int)g);
                if (g < v.phasing.size() && !v.phasing[g]) px.any_unphased[j] = 1;
            }
        }
        return px;
    }

    // build_alleles_combs + combine_haplotypes, var_block.hpp:709-786.  A set of allele picks along the
    // chain; picks are compared by allele TEXT in the reference, i.e. by first index with that text.
    std::set<std::vector<int>> allele_picks(const std::vector<int> &comb, int central, bool haploid, const PanelIndex *px = nullptr) const
    {
        std::set<std::vector<int>> out;
        auto canon = [&](int var, int allele) {
            const Variant &v = vars[(size_t)var];
            return v.allele_index(v.allele(allele)); // .at() throws if the GT names a dropped symbolic allele
        };
        const size_t n = comb.size();
        // the samples to walk: all of them, or (sparse form) those non-reference somewhere on the chain
        std::vector<int> walk;
        bool sparse = px != nullptr;
        if (sparse)
            for (int m : comb) sparse = sparse && px->regular[(size_t)m];
        if (sparse) {
            bool unphased_somewhere = false;
            for (int m : comb) {
                walk.insert(walk.end(), px->nonref[(size_t)m].begin(), px->nonref[(size_t)m].end());
                unphased_somewhere = unphased_somewhere || px->any_unphased[(size_t)m];
            }
            std::sort(walk.begin(), walk.end());
            walk.erase(std::unique(walk.begin(), walk.end()), walk.end());
            // the dense walk throws on the first unphased sample of a chain longer than 24, all-reference or not
            if (!haploid && n > 24 && unphased_somewhere) throw std::runtime_error("unphased chain of more than 24 variants (2^n haplotypes)");
            if (walk.size() < px->n_samples) out.insert(std::vector<int>(n, 0)); // some sample is all-reference here
        }
        const size_t n_walk = sparse ? walk.size() : vars[(size_t)central].genotypes.size();
        for (size_t w = 0; w < n_walk; ++w) {
            const size_t g = sparse ? (size_t)walk[w] : w;
            std::vector<int> h1(n), h2(n);
            bool phased = true;
            for (size_t j = 0; j < n; ++j) {
                const Variant &v = vars[(size_t)comb[j]];
                h1[j] = canon(comb[j], v.genotypes.at(g).first);
                if (!haploid) {
                    h2[j] = canon(comb[j], v.genotypes.at(g).second);
                    phased = phased && v.phasing.at(g);
                }
            }
            if (haploid) {
                out.insert(h1);
            } else if (phased) {
                out.insert(h1);
                out.insert(h2);
            } else {
                if (n > 24) throw std::runtime_error("unphased chain of more than 24 variants (2^n haplotypes)");
                for (uint32_t m = 0; m < (1u << n); ++m) { // every pick of hap1/hap2 per level
                    std::vector<int> h(n);
                    for (size_t j = 0; j < n; ++j) h[j] = (m >> j) & 1 ? h2[j] : h1[j];
                    out.insert(std::move(h));
                }
            }
        }
        return out;
    }

    // extract() of a block for which is_lone_short() holds, without its containers: the only chain is the variant
    // itself, each allele carried by a panel haplotype has one signature of one k-mer (the allele centred at k/2
    // between its reference flanks); emit(allele, kmer) once per such allele.  Same clipping and the same
    // exceptions as the general form.
    template <class F> void extract_lone(const std::string &reference, bool haploid, F emit) const
    {
        const Variant &v = vars[0];
        if (!v.is_present || v.ref_pos < k || v.ref_pos > (int)reference.size() - k) return;
        uint64_t carried = 0;
        for (size_t g = 0; g < v.genotypes.size(); ++g) { // build_alleles_combs on a chain of one
            carried |= 1ULL << v.allele_index(v.allele(v.genotypes[g].first));
            if (!haploid) {
                carried |= 1ULL << v.allele_index(v.allele(v.genotypes[g].second));
                (void)v.phasing.at(g);
            }
        }
        for (int a = 0; a < v.n_alleles(); ++a) {
            if (!((carried >> a) & 1)) continue;
            const std::string &al = v.allele(a);
            const int alen = (int)al.size();
            const int missing_prefix = k / 2 - alen / 2, missing_suffix = (k + 1) / 2 - (alen - alen / 2);
            std::string kmer = substr_clip(reference, (long)v.ref_pos - missing_prefix, missing_prefix);
            kmer += al;
            kmer += substr_clip(reference, (long)v.ref_pos + v.ref_size, missing_suffix);
            emit(a, kmer);
        }
    }
    // every block: the fast form where it applies
    std::vector<AlleleSignatures> signatures(const std::string &reference, bool haploid) const
    {
        if (!is_lone_short()) return extract(reference, haploid);
        std::vector<AlleleSignatures> result(1);
        extract_lone(reference, haploid, [&](int a, const std::string &kmer) { result[0][a].push_back({kmer}); });
        return result;
    }

    // extract_kmers, var_block.hpp:95-219
    std::vector<AlleleSignatures> extract(const std::string &reference, bool haploid) const
    {
        std::vector<AlleleSignatures> result(vars.size());
        const PanelIndex panel = panel_index();
        // the variants of a block are enumerated independently of one another: a large block (C1 has one of 8,724
        // variants) is cut over the host cores, interleaved because chain counts vary along the block
        const int n_threads = vars.size() < 64 ? 1 : (int)std::max(1u, std::min(std::thread::hardware_concurrency(), 16u));
        if (n_threads > 1) {
            std::vector<std::exception_ptr> errs((size_t)n_threads);
            std::vector<int> err_at((size_t)n_threads, INT_MAX);
            std::vector<std::thread> pool;
            for (int t = 0; t < n_threads; ++t)
                pool.emplace_back([&, t]() {
                    for (int vi = t; vi < (int)vars.size(); vi += n_threads) {
                        try {
                            extract_one(vi, reference, haploid, panel, result[(size_t)vi]);
                        } catch (...) {
                            errs[(size_t)t] = std::current_exception();
                            err_at[(size_t)t] = vi;
                            return;
                        }
                    }
                });
            for (auto &th : pool) th.join();
            int first = -1;
            for (int t = 0; t < n_threads; ++t)
                if (errs[(size_t)t] && (first < 0 || err_at[(size_t)t] < err_at[(size_t)first])) first = t;
            if (first >= 0) std::rethrow_exception(errs[(size_t)first]); // the one the sequential walk would have met first
            return result;
        }
        for (int vi = 0; vi < (int)vars.size(); ++vi) extract_one(vi, reference, haploid, panel, result[(size_t)vi]);
        return result;
    }

    // the signatures of variant vi (one element of extract()'s result)
    void extract_one(int vi, const std::string &reference, bool haploid, const PanelIndex &panel, AlleleSignatures &out) const
    {
        {
            const Variant &v = vars[(size_t)vi];
            if (!v.is_present || v.ref_pos < k || v.ref_pos > (int)reference.size() - k) return;
            const auto combs = combine(chains(vi, -1, &panel), chains(vi, +1, &panel), vi);
            for (const auto &comb : combs) {
                // get_ref_subs, var_block.hpp:682-702
                std::vector<std::string> rsubs;
                int last_end = -1;
                for (int index : comb) {
                    const Variant &cv = vars[(size_t)index];
                    if (last_end != -1) rsubs.push_back(substr_clip(reference, last_end, cv.ref_pos - last_end));
                    last_end = cv.ref_pos + cv.ref_size;
                }
                for (const auto &pick : allele_picks(comb, vi, haploid, &panel)) {
                    std::vector<std::string> sig;
                    std::string mid_allele;
                    if (pick.size() == 1 && (int)vars[(size_t)comb[0]].allele(pick[0]).size() >= k) {
                        mid_allele = vars[(size_t)comb[0]].allele(pick[0]);
                        for (size_t p = 0; p + (size_t)k <= mid_allele.size(); ++p) sig.push_back(mid_allele.substr(p, (size_t)k));
                    } else {
                        std::string kmer;
                        int mid_pos = 0;
                        for (size_t j = 0; j < pick.size(); ++j) {
                            const std::string &al = vars[(size_t)comb[j]].allele(pick[j]);
                            if (comb[j] == vi) {
                                mid_pos = (int)kmer.size();
                                mid_allele = al;
                            }
                            kmer += al;
                            if (j < rsubs.size()) kmer += rsubs[j];
                        }
                        const int first_part = mid_pos + (int)mid_allele.size() / 2;
                        const int second_part = (int)kmer.size() - first_part;
                        const int missing_prefix = k / 2 - first_part;
                        const int missing_suffix = (k + 1) / 2 - second_part;
                        if (missing_prefix >= 0) {
                            const Variant &fv = vars[(size_t)comb.front()];
                            kmer = substr_clip(reference, (long)fv.ref_pos - missing_prefix, missing_prefix) + kmer;
                        } else
                            kmer.erase(0, (size_t)-missing_prefix);
                        if (missing_suffix >= 0) {
                            const Variant &lv = vars[(size_t)comb.back()];
                            kmer += substr_clip(reference, (long)lv.ref_pos + lv.ref_size, missing_suffix);
                        } else {
                            if ((size_t)-missing_suffix > kmer.size()) throw std::out_of_range("k-mer shorter than the cut");
                            kmer.erase(kmer.size() - (size_t)-missing_suffix);
                        }
                        sig.push_back(std::move(kmer));
                    }
                    out[v.allele_index(mid_allele)].push_back(std::move(sig));
                }
            }
        }
    }
};

} // namespace malva
