// vcf_format.cpp — SURVEY.md section 8 row f3: the VCF body line of a called allele straight from the 64-byte record
// (src/lib/Pisces.IO/VcfFileWriter.cs:206-262 WriteListOfColocatedAlleles, VcfFormatter.cs:52-495): one line per allele
// (AllowMultipleVcfLinesPerLoci, the Pisces default) or one per position (crushed), with RegionMapper's no-call padding of uncovered
// interval positions (RegionMapper.cs:31-84).  Pure host code, no device.
//
// Number formatting follows .NET Core 2.0, which the reference targets: Single.ToString("0.000") first takes the 7 significant
// decimal digits of the float (15 for Double), then rounds that digit string half-up to the requested decimals, and a negative
// value that rounds to zero prints without the sign.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "../../include/pisces_hip.h"

namespace {

// value rounded to `sig` significant digits (correctly, on the exact binary value), then half-up to `decimals`
std::string dotnet_fixed(double value, int sig, int decimals)
{
    if (std::isnan(value)) return "NaN";
    if (std::isinf(value)) return value > 0 ? "Infinity" : "-Infinity";
    char buf[64];
    std::snprintf(buf, sizeof buf, "%.*e", sig - 1, std::fabs(value));   // d.ddddddde[+-]xx
    std::string digits;
    int exp10 = 0;
    {
        const char* e = std::strchr(buf, 'e');
        for (const char* p = buf; p < e; p++)
            if (*p >= '0' && *p <= '9') digits.push_back(*p);
        exp10 = std::atoi(e + 1);
    }
    // the number is 0.DIGITS x 10^(exp10 + 1); keep `decimals` digits after the point
    const int point = exp10 + 1;                       // digits before the decimal point (may be <= 0)
    int keep = point + decimals;                       // digits of `digits` that survive
    std::string kept;
    bool round_up = false;
    if (keep < 0) {
        kept.clear();
    } else {
        if ((size_t)keep < digits.size()) {
            round_up = digits[(size_t)keep] >= '5';
            kept = digits.substr(0, (size_t)keep);
        } else {
            kept = digits + std::string((size_t)keep - digits.size(), '0');
        }
    }
    int point_in_kept = point;
    if (round_up) {
        int i = (int)kept.size() - 1;
        while (i >= 0 && kept[(size_t)i] == '9') { kept[(size_t)i] = '0'; i--; }
        if (i >= 0) kept[(size_t)i]++;
        else { kept.insert(kept.begin(), '1'); point_in_kept++; }
    }
    // assemble integer part / fraction part
    std::string ip, fp;
    if (point_in_kept <= 0) {
        ip = "0";
        fp = std::string((size_t)(-point_in_kept), '0') + kept;
    } else {
        ip = kept.substr(0, (size_t)point_in_kept);
        fp = kept.substr((size_t)point_in_kept);
    }
    if ((int)fp.size() < decimals) fp += std::string((size_t)decimals - fp.size(), '0');
    fp = fp.substr(0, (size_t)decimals);
    while (ip.size() > 1 && ip[0] == '0') ip.erase(ip.begin());
    bool all_zero = true;
    for (char c : ip + fp) all_zero &= (c == '0');
    std::string out = (value < 0 && !all_zero) ? "-" : "";
    out += ip;
    if (decimals > 0) out += "." + fp;
    return out;
}

std::string fmt_single(float v, int decimals) { return dotnet_fixed((double)v, 7, decimals); }
std::string fmt_double(double v, int decimals) { return dotnet_fixed(v, 15, decimals); }

// VcfFormatter.GetNumSigDigits (:66-71) on Single.ToString() of the threshold
int sig_digits_of(float threshold)
{
    char buf[64];
    std::snprintf(buf, sizeof buf, "%.7g", (double)threshold);   // Single.ToString(): up to 7 significant digits, shortest form
    std::string s(buf);
    const size_t e = s.find('e');
    if (e != std::string::npos) return std::abs(std::atoi(s.c_str() + e + 1));   // "1E-05" -> 5
    return (int)s.size() - 1;
}

const char* map_genotype(int gt)   // VcfFormatter.MapGenotype :184-216
{
    switch (gt) {
    case PISCES_GT_HOM_ALT: return "1/1";
    case PISCES_GT_HOM_REF: return "0/0";
    case PISCES_GT_HET_ALT_REF: return "0/1";
    case PISCES_GT_HET_ALT1_ALT2: return "1/2";
    case PISCES_GT_REF_LIKE_NOCALL: return "./.";
    case PISCES_GT_ALT_LIKE_NOCALL: return "./.";
    case PISCES_GT_REF_AND_NOCALL: return "0/.";
    case PISCES_GT_ALT_AND_NOCALL: return "1/.";
    case PISCES_GT_HEMI_ALT: return "1";
    case PISCES_GT_HEMI_NOCALL: return ".";
    case PISCES_GT_HEMI_REF: return "0";
    case PISCES_GT_OTHERS: return "2/2";
    default: return "./.";
    }
}

}  // namespace

extern "C" {

int32_t pisces_hip_vcf_default_config(PiscesVcfConfig* c)
{
    if (!c) return PISCES_E_INVALID_ARG;
    std::memset(c, 0, sizeof(*c));
    c->variant_quality_filter = 30;
    c->rmxn_max_repeat_length = 5;
    c->rmxn_min_repetitions = 9;
    c->noise_level = 20;
    c->output_strand_bias_and_noise_level = 1;
    c->noise_level_from_records = 0;
    c->output_no_call_fraction = 0;
    c->min_frequency_threshold = 0.01f;
    c->frequency_filter_threshold = 0.01f;
    c->crush = 0;
    return PISCES_OK;
}

int64_t pisces_hip_format_vcf_padded(const PiscesVcfConfig* cfg, const char* chrom, const PiscesCalledAllele* recs, int64_t n,
                                     const int32_t* cand_index, const PiscesCandidate* cands, const uint8_t* alleles,
                                     const uint8_t* ref_bases, int64_t ref_len, const int32_t* interval_starts,
                                     const int32_t* interval_ends, int32_t n_intervals, PiscesVcfPadState* state, int32_t finish,
                                     char* out, int64_t capacity)
{
    try {   // nothing crosses the C ABI as an exception (std::bad_alloc from the text buffer, ...)
    return [&]() -> int64_t {
    if (!cfg || !chrom || n < 0 || (n > 0 && !recs) || capacity < 0 || (capacity > 0 && !out)) return PISCES_E_INVALID_ARG;
    const bool pad = state != nullptr;
    if (pad && (n_intervals < 0 || (n_intervals > 0 && (!interval_starts || !interval_ends)) || !ref_bases || ref_len < 0)) return PISCES_E_INVALID_ARG;
    // VcfFormatter.UpdateFrequencyFormat :52-64
    int freq_decimals = sig_digits_of(cfg->min_frequency_threshold);
    if (cfg->frequency_filter_threshold >= 0.0f) freq_decimals = std::max(freq_decimals, sig_digits_of(cfg->frequency_filter_threshold));
    static const char kBase[6] = {'A', 'G', 'C', 'T', 'N', 'D'};
    std::string format = "GT:GQ:AD:DP:VF";
    if (cfg->output_strand_bias_and_noise_level) format += ":NL:SB";
    if (cfg->output_no_call_fraction) format += ":NC";
    std::string text;
    PiscesVcfPadState st = pad ? *state : PiscesVcfPadState{0, 0, -1};

    // RegionMapper.GetNextEmptyCall :31-53 (+ GetNextRegion :55-68): next interval position in [startPosition, maxUpTo] not yet padded
    int32_t interval_max = 0;
    for (int32_t i = 0; i < n_intervals; i++) interval_max = std::max(interval_max, interval_ends[i]);
    auto next_empty_call = [&](int32_t startPosition, bool has_max, int32_t maxUpToPosition, int32_t& position) {
        int32_t region = -1;
        for (int32_t i = st.last_cleared_interval_index + 1; i < n_intervals; i++) {
            if (interval_ends[i] >= startPosition) { region = i; break; }
            st.last_cleared_interval_index++;
        }
        if (region < 0) return false;
        const int32_t nextPosition = std::max(interval_starts[region], std::max(st.last_padded_position + 1, startPosition));
        const int32_t endPosition = has_max ? std::min(maxUpToPosition, interval_max) : interval_max;
        if (nextPosition > endPosition) return false;
        if (interval_ends[region] <= nextPosition) st.last_cleared_interval_index++;
        if (!(nextPosition >= interval_starts[region] && nextPosition <= interval_ends[region])) return false;
        st.last_padded_position = nextPosition;
        position = nextPosition;
        return true;
    };
    // RegionMapper.GetMissingReference :70-84 through WriteListOfColocatedAlleles: a Reference allele with nothing but the noise level
    auto write_no_call = [&](int32_t position) {
        const char refBase = (position >= 1 && position <= ref_len) ? (char)ref_bases[position - 1] : 'N';
        std::string sample = "./.:0:0:0:" + fmt_single(0.0f, freq_decimals);
        if (cfg->output_strand_bias_and_noise_level) sample += ":" + std::to_string(cfg->noise_level) + ":" + fmt_double(0.0, 4);
        if (cfg->output_no_call_fraction) sample += ":" + fmt_single(0.0f, 4);
        text += chrom;
        text += "\t" + std::to_string(position) + "\t.\t" + std::string(1, refBase) + "\t.\t0\tLowDP\tDP=0\t" + format + "\t" + sample + "\n";
        st.last_variant_position_written = position;
    };
    auto pad_if_needed = [&](int32_t position) {   // VcfFileWriter.PadIfNeeded :124-140
        if (!pad) return;
        if (st.last_variant_position_written == 0 || st.last_variant_position_written + 1 < position) {
            int32_t p;
            while (next_empty_call(st.last_variant_position_written + 1, true, position - 1, p)) write_no_call(p);
        }
    };

    auto allele_strings = [&](int64_t i, std::string& ref_allele, std::string& alt_allele) -> bool {
        const PiscesCalledAllele& r = recs[i];
        if (cand_index && cand_index[i] >= 0) {
            if (!cands || !alleles) return false;
            const PiscesCandidate& c = cands[cand_index[i]];
            ref_allele.assign((const char*)alleles + c.allele_offset, (size_t)c.ref_len);
            alt_allele.assign((const char*)alleles + c.allele_offset + c.ref_len, (size_t)c.alt_len);
        } else {
            ref_allele.assign(1, kBase[PISCES_INFO_REF(r.info)]);
            alt_allele.assign(1, kBase[PISCES_INFO_ALT(r.info)]);
        }
        return true;
    };
    auto frequency_of = [](const PiscesCalledAllele& r) {   // CalledAllele.Frequency (CalledAllele.cs:49-52)
        float freq = r.total_coverage == 0 ? 0.0f : (float)r.allele_support / (float)r.total_coverage;
        return freq > 1.0f ? 1.0f : freq;
    };

    for (int64_t g0 = 0; g0 < n;) {
        // GroupsAllelesThenWrite :174-204: co-located alleles share a line when crushing; otherwise one allele per line
        int64_t g1 = g0 + 1;
        if (cfg->crush)
            while (g1 < n && recs[g1].position == recs[g0].position) g1++;
        const PiscesCalledAllele& first = recs[g0];
        pad_if_needed(first.position);
        const int gt = PISCES_INFO_GENOTYPE(first.info);
        const bool is_ref = PISCES_INFO_CATEGORY(first.info) == PISCES_CAT_REFERENCE;
        const bool alt12 = gt == PISCES_GT_HET_ALT1_ALT2 || gt == PISCES_GT_ALT12_LIKE_NOCALL;
        const bool others = gt == PISCES_GT_OTHERS;   // a forced allele beside the called ones (DiploidLocusProcessor.cs:36-38)
        const bool forced_to_report = (first.filter_bits >> PISCES_FILTER_FORCED_REPORT) & 1u;   // CalledAllele.IsForcedToReport
        // GetDepthCountInt :373-394
        int depth = is_ref ? first.reference_support : first.reference_support + first.allele_support;
        int total_variant_reads = 0, qual = first.variant_qscore, gq = first.genotype_qscore;
        for (int64_t i = g0; i < g1; i++) {
            depth = std::max(depth, recs[i].total_coverage);
            total_variant_reads += recs[i].allele_support;
            qual = std::min(qual, recs[i].variant_qscore);      // MergeVariantQScores :486-489
            gq = std::min(gq, (int)recs[i].genotype_qscore);         // MergeGenotypeQScores :491-494
        }
        depth = std::max(depth, total_variant_reads);
        // SetUncrushedReferenceAndAlt :434-448 (PhaseSetIndex: 1 / 2 for the variant alleles of a diploid call, 0 on the somatic path) /
        // MergeCrushedReferenceAndAlt :450-482
        const int phase = PISCES_FILTERBITS_PHASE(first.filter_bits);
        std::string ref_allele, alt_allele;
        if (g1 - g0 == 1) {
            if (!allele_strings(g0, ref_allele, alt_allele)) return PISCES_E_INVALID_ARG;
            if (alt12 || others) alt_allele = (phase == 1 || others) ? alt_allele + ",<M>" : "<M>," + alt_allele;
        } else {
            std::vector<std::pair<std::string, std::string>> ra((size_t)(g1 - g0));
            for (int64_t i = g0; i < g1; i++) {
                if (!allele_strings(i, ra[(size_t)(i - g0)].first, ra[(size_t)(i - g0)].second)) return PISCES_E_INVALID_ARG;
                if (ra[(size_t)(i - g0)].first.size() > ref_allele.size()) ref_allele = ra[(size_t)(i - g0)].first;
            }
            for (size_t k = 0; k < ra.size(); k++) {
                std::string rep = ra[k].second;
                if (ref_allele.size() != ra[k].first.size()) rep += ref_allele.substr(ra[k].first.size());
                if (k) alt_allele += ",";
                alt_allele += rep;
            }
        }
        const bool ref_like_gt = gt == PISCES_GT_HOM_REF || gt == PISCES_GT_REF_LIKE_NOCALL || gt == PISCES_GT_REF_AND_NOCALL ||
                                 gt == PISCES_GT_HEMI_NOCALL || gt == PISCES_GT_HEMI_REF;   // VcfFileWriter.cs:236-240
        // MapFilters / MapFilter :136-182 over MergeFilters :423-432: each allele's filters in the order AlleleProcessor.ApplyFilters adds
        // them (src/exe/Pisces/Logic/VariantCalling/AlleleProcessor.cs:25-71), alleles in order, first occurrence kept
        std::string filters;
        uint32_t seen = 0;
        // (MultiAllelicSite is added by the diploid genotyper, GenotypeCalculatorUtilities.cs:139-145, after the processor's filters
        // and before AlleleCaller's LowGQ; ForcedReport by AlleleCaller.cs:112-116, after the processor's filters and before the genotyper)
        static const int kOrder[9] = {PISCES_FILTER_LOW_DEPTH, PISCES_FILTER_LOW_VARIANT_QSCORE, PISCES_FILTER_NO_CALL, PISCES_FILTER_STRAND_BIAS,
                                      PISCES_FILTER_RMXN, PISCES_FILTER_LOW_VARIANT_FREQUENCY, PISCES_FILTER_FORCED_REPORT,
                                      PISCES_FILTER_MULTI_ALLELIC_SITE, PISCES_FILTER_LOW_GENOTYPE_QUALITY};
        for (int64_t i = g0; i < g1; i++)
            for (int f : kOrder) {
                if (!(recs[i].filter_bits & (1u << f)) || (seen & (1u << f))) continue;
                seen |= 1u << f;
                std::string name;
                switch (f) {
                case PISCES_FILTER_LOW_DEPTH: name = "LowDP"; break;
                case PISCES_FILTER_LOW_VARIANT_QSCORE:
                    if (cfg->variant_quality_filter < 0) return PISCES_E_INVALID_ARG;   // InvalidDataException in the reference
                    name = "q" + std::to_string(cfg->variant_quality_filter);
                    break;
                case PISCES_FILTER_NO_CALL: name = "NC"; break;
                case PISCES_FILTER_STRAND_BIAS: name = "SB"; break;
                case PISCES_FILTER_RMXN:
                    if (cfg->rmxn_max_repeat_length < 0 || cfg->rmxn_min_repetitions < 0) return PISCES_E_INVALID_ARG;
                    name = "R" + std::to_string(cfg->rmxn_max_repeat_length) + "x" + std::to_string(cfg->rmxn_min_repetitions);
                    break;
                case PISCES_FILTER_LOW_VARIANT_FREQUENCY: name = "LowVariantFreq"; break;
                case PISCES_FILTER_FORCED_REPORT: name = "ForcedReport"; break;
                case PISCES_FILTER_MULTI_ALLELIC_SITE: name = "MultiAllelicSite"; break;
                default: name = "LowGQ"; break;
                }
                if (!filters.empty()) filters += ";";
                filters += name;
            }
        if (filters.empty()) filters = "PASS";
        // GetAlleleCountString :396-421, GetFrequencyString :329-358
        std::string ad, vf;
        const float freq = frequency_of(first);
        if (is_ref) {
            ad = std::to_string(first.allele_support);
            vf = first.total_coverage == 0 ? fmt_single(0.0f, freq_decimals) : fmt_single(1.0f - freq, freq_decimals);
        } else if (alt12 || others) {
            if (g1 - g0 > 1) {
                for (int64_t i = g0; i < g1; i++) ad += (i > g0 ? "," : "") + std::to_string(recs[i].allele_support);
            } else {
                const int other = depth - first.allele_support - first.reference_support;
                ad = (phase == 1 || others) ? std::to_string(first.reference_support) + "," + std::to_string(first.allele_support) + "," + std::to_string(other)
                                : std::to_string(first.reference_support) + "," + std::to_string(other) + "," + std::to_string(first.allele_support);
            }
            if (others) {   // GetFrequencyString :341 names the 1/2 genotypes only
                vf = fmt_single(freq, freq_decimals);
            } else {
                double sum = 0.0;   // SumMultipleVF
                for (int64_t i = g0; i < g1; i++) sum += (double)recs[i].allele_support / (double)depth;
                vf = fmt_double(sum, freq_decimals);
            }
        } else {
            ad = std::to_string(first.reference_support) + "," + std::to_string(first.allele_support);
            vf = fmt_single(freq, freq_decimals);
        }
        std::string sample = std::string(map_genotype(gt)) + ":" + std::to_string(gq) + ":" + ad + ":" + std::to_string(depth) + ":" + vf;
        if (cfg->output_strand_bias_and_noise_level) {
            // NoiseLevelApplied is set where the q-score is computed (VariantQualityCalculator.cs:13), i.e. for support > 0;
            // BiasResults.GATKBiasScore = 10 log10(BiasScore) (MathOperations.PtoGATKBiasScale), default 0 when never computed
            const long long nl = cfg->noise_level_from_records ? (first.noise_level == INT16_MIN ? (long long)INT32_MIN : (long long)first.noise_level)
                                                               : (first.allele_support > 0 ? cfg->noise_level : 0);
            double gatk = first.allele_support > 0 ? 10.0 * std::log10(first.strand_bias_score) : 0.0;
            gatk = std::min(std::max(-100.0, gatk), 0.0);
            sample += ":" + std::to_string(nl) + ":" + fmt_double(gatk, 4);
        }
        if (cfg->output_no_call_fraction) {
            const float all = (float)(first.total_coverage + first.num_no_calls);   // CalledAllele.SetFractionNoCalls :107-114
            const float nc = all == 0.0f ? 0.0f : (float)first.num_no_calls / all;
            sample += ":" + fmt_single(nc, 4);
        }
        text += chrom;
        text += "\t" + std::to_string(first.position) + "\t.\t" + ref_allele + "\t" + ((ref_like_gt && !forced_to_report) ? std::string(".") : alt_allele) + "\t" +
                std::to_string(qual) + "\t" + filters + "\tDP=" + std::to_string(depth) + "\t" + format + "\t" + sample + "\n";
        st.last_variant_position_written = first.position;
        g0 = g1;
    }
    if (pad && finish) {   // WriteRemaining :149-165
        int32_t p;
        while (next_empty_call(st.last_variant_position_written + 1, false, 0, p)) write_no_call(p);
        st.last_variant_position_written = 0;
    }
    const int64_t need = (int64_t)text.size();
    if (need <= capacity) {
        if (need > 0) std::memcpy(out, text.data(), (size_t)need);
        if (pad) *state = st;
    }
    return need;   // > capacity: nothing was written and the state is unchanged, call again with a buffer of this size
    }();
    } catch (...) {
        return PISCES_E_INTERNAL;
    }
}

int64_t pisces_hip_format_vcf(const PiscesVcfConfig* cfg, const char* chrom, const PiscesCalledAllele* recs, int64_t n,
                              const int32_t* cand_index, const PiscesCandidate* cands, const uint8_t* alleles, char* out,
                              int64_t capacity)
{
    return pisces_hip_format_vcf_padded(cfg, chrom, recs, n, cand_index, cands, alleles, nullptr, 0, nullptr, nullptr, 0, nullptr, 0, out, capacity);
}

}  // extern "C"
