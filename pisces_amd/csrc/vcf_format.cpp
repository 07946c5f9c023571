// vcf_format.cpp — SURVEY.md section 8 row f3: the VCF body line of a called allele straight from the 64-byte record
// (src/lib/Pisces.IO/VcfFileWriter.cs:206-262 WriteListOfColocatedAlleles, VcfFormatter.cs:52-448), uncrushed form
// (AllowMultipleVcfLinesPerLoci, the Pisces default: one line per allele).  Pure host code, no device.
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
    c->output_no_call_fraction = 0;
    c->min_frequency_threshold = 0.01f;
    c->frequency_filter_threshold = 0.01f;
    return PISCES_OK;
}

int64_t pisces_hip_format_vcf(const PiscesVcfConfig* cfg, const char* chrom, const PiscesCalledAllele* recs, int64_t n,
                              const int32_t* cand_index, const PiscesCandidate* cands, const uint8_t* alleles, char* out,
                              int64_t capacity)
{
    if (!cfg || !chrom || n < 0 || (n > 0 && !recs) || capacity < 0 || (capacity > 0 && !out)) return PISCES_E_INVALID_ARG;
    // VcfFormatter.UpdateFrequencyFormat :52-64
    int freq_decimals = sig_digits_of(cfg->min_frequency_threshold);
    if (cfg->frequency_filter_threshold >= 0.0f) freq_decimals = std::max(freq_decimals, sig_digits_of(cfg->frequency_filter_threshold));
    static const char kBase[6] = {'A', 'G', 'C', 'T', 'N', 'D'};
    std::string text;
    for (int64_t i = 0; i < n; i++) {
        const PiscesCalledAllele& r = recs[i];
        const int gt = PISCES_INFO_GENOTYPE(r.info), cat = PISCES_INFO_CATEGORY(r.info);
        const bool is_ref = cat == PISCES_CAT_REFERENCE;
        std::string ref_allele, alt_allele;
        if (cand_index && cand_index[i] >= 0) {
            if (!cands || !alleles) return PISCES_E_INVALID_ARG;
            const PiscesCandidate& c = cands[cand_index[i]];
            ref_allele.assign((const char*)alleles + c.allele_offset, (size_t)c.ref_len);
            alt_allele.assign((const char*)alleles + c.allele_offset + c.ref_len, (size_t)c.alt_len);
        } else {
            ref_allele.assign(1, kBase[PISCES_INFO_REF(r.info)]);
            alt_allele.assign(1, kBase[PISCES_INFO_ALT(r.info)]);
        }
        // GetDepthCountInt :373-394 for a single allele
        int depth = is_ref ? r.reference_support : r.reference_support + r.allele_support;
        depth = std::max(depth, r.total_coverage);
        depth = std::max(depth, r.allele_support);
        // CalledAllele.Frequency (CalledAllele.cs:49-52)
        float freq = r.total_coverage == 0 ? 0.0f : (float)r.allele_support / (float)r.total_coverage;
        if (freq > 1.0f) freq = 1.0f;
        // SetUncrushedReferenceAndAlt :434-448 (PhaseSetIndex is 0 on this path)
        if (gt == PISCES_GT_HET_ALT1_ALT2 || gt == PISCES_GT_ALT12_LIKE_NOCALL) alt_allele = "<M>," + alt_allele;
        const bool ref_like_gt = gt == PISCES_GT_HOM_REF || gt == PISCES_GT_REF_LIKE_NOCALL || gt == PISCES_GT_REF_AND_NOCALL;
        // MapFilters / MapFilter :136-182, in the order AlleleProcessor.ApplyFilters adds them (src/exe/Pisces/Logic/VariantCalling/
        // AlleleProcessor.cs:25-71)
        std::string filters;
        auto add = [&](const std::string& f) { if (!filters.empty()) filters += ";"; filters += f; };
        const uint32_t fb = r.filter_bits;
        if (fb & (1u << PISCES_FILTER_LOW_DEPTH)) add("LowDP");
        if (fb & (1u << PISCES_FILTER_LOW_VARIANT_QSCORE)) {
            if (cfg->variant_quality_filter < 0) return PISCES_E_INVALID_ARG;   // InvalidDataException in the reference
            add("q" + std::to_string(cfg->variant_quality_filter));
        }
        if (fb & (1u << PISCES_FILTER_NO_CALL)) add("NC");
        if (fb & (1u << PISCES_FILTER_STRAND_BIAS)) add("SB");
        if (fb & (1u << PISCES_FILTER_RMXN)) {
            if (cfg->rmxn_max_repeat_length < 0 || cfg->rmxn_min_repetitions < 0) return PISCES_E_INVALID_ARG;
            add("R" + std::to_string(cfg->rmxn_max_repeat_length) + "x" + std::to_string(cfg->rmxn_min_repetitions));
        }
        if (fb & (1u << PISCES_FILTER_LOW_VARIANT_FREQUENCY)) add("LowVariantFreq");
        if (fb & (1u << PISCES_FILTER_LOW_GENOTYPE_QUALITY)) add("LowGQ");
        if (filters.empty()) filters = "PASS";
        // GetAlleleCountString :396-421, GetFrequencyString :329-358
        std::string ad, vf;
        if (is_ref) {
            ad = std::to_string(r.allele_support);
            vf = r.total_coverage == 0 ? fmt_single(0.0f, freq_decimals) : fmt_single(1.0f - freq, freq_decimals);
        } else if (gt == PISCES_GT_HET_ALT1_ALT2 || gt == PISCES_GT_ALT12_LIKE_NOCALL) {
            const int other = depth - r.allele_support - r.reference_support;
            ad = std::to_string(r.reference_support) + "," + std::to_string(other) + "," + std::to_string(r.allele_support);
            vf = fmt_double((double)r.allele_support / (double)depth, freq_decimals);
        } else {
            ad = std::to_string(r.reference_support) + "," + std::to_string(r.allele_support);
            vf = fmt_single(freq, freq_decimals);
        }
        std::string format = "GT:GQ:AD:DP:VF", sample = std::string(map_genotype(gt)) + ":" + std::to_string(r.genotype_qscore) + ":" + ad +
                                                        ":" + std::to_string(depth) + ":" + vf;
        if (cfg->output_strand_bias_and_noise_level) {
            // NoiseLevelApplied is set where the q-score is computed (VariantQualityCalculator.cs:13), i.e. for support > 0;
            // BiasResults.GATKBiasScore = 10 log10(BiasScore) (MathOperations.PtoGATKBiasScale), default 0 when never computed
            const int nl = r.allele_support > 0 ? cfg->noise_level : 0;
            double gatk = r.allele_support > 0 ? 10.0 * std::log10(r.strand_bias_score) : 0.0;
            gatk = std::min(std::max(-100.0, gatk), 0.0);
            format += ":NL:SB";
            sample += ":" + std::to_string(nl) + ":" + fmt_double(gatk, 4);
        }
        if (cfg->output_no_call_fraction) {
            const float all = (float)(r.total_coverage + r.num_no_calls);   // CalledAllele.SetFractionNoCalls :107-114
            const float nc = all == 0.0f ? 0.0f : (float)r.num_no_calls / all;
            format += ":NC";
            sample += ":" + fmt_single(nc, 4);
        }
        text += chrom;
        text += "\t" + std::to_string(r.position) + "\t.\t" + ref_allele + "\t" + (ref_like_gt ? std::string(".") : alt_allele) + "\t" +
                std::to_string(r.variant_qscore) + "\t" + filters + "\tDP=" + std::to_string(depth) + "\t" + format + "\t" + sample + "\n";
    }
    const int64_t need = (int64_t)text.size();
    if (need <= capacity && need > 0) std::memcpy(out, text.data(), (size_t)need);
    return need;   // > capacity: nothing was written, call again with a buffer of this size
}

}  // extern "C"
