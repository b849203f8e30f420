/*
 * sqy_oracle_float.c -- the float-bearing parts of the CPU restatement (quantiser LUT
 * construction, frame_shuffle metric).  TEST INFRASTRUCTURE ONLY (see sqy_oracle.h).
 *
 * Must be compiled without -ffast-math and with -ffp-contract=off: the declared parity target is
 * IEEE-754 binary32/binary64 evaluation in the reference's statement order (its release build uses
 * -Ofast, which makes these paths compiler dependent -- SURVEY.md F10).
 *
 * Paths cited are relative to /root/reference/src/cpp/src.
 */
#include "sqy_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* encoders/histogram_utils.hpp:39-55 (serial::fill_histogram) */
void sqo_histogram_u16(const uint16_t* in, size_t len, uint32_t* histo)
{
    memset(histo, 0, 65536 * sizeof(uint32_t));
    for (size_t i = 0; i < len; ++i) histo[in[i]] += 1;
}

void sqo_histogram_u8(const uint8_t* in, size_t len, uint32_t* histo)
{
    memset(histo, 0, 256 * sizeof(uint32_t));
    for (size_t i = 0; i < len; ++i) histo[in[i]] += 1;
}

/* encoders/quantiser_weighters.hpp:20-160 as selected by quantiser_scheme_impl.hpp:186-198 (computeWeights,
 * quantiser_utils.hpp:317-322; the weights start out as 1.f, :85,104):
 *   mode 0  none                       weights stay 1.f
 *   mode 1  power_of(num, den)         weights[i] = std::pow(i, exponent) for every bin             (:121-136)
 *   mode 2  offset_power_of(num, den)  weights[i] = std::pow(i - offset, exponent) for i >= offset, offset = first non-zero
 *                                      bin of the histogram; the bins below keep 1.f               (:40-84)
 * exponent = float(num) / den (:32,110).  i is an integer (omp_size_type), so std::pow promotes both arguments to double
 * and the result is narrowed to the float element. */
void sqo_quantiser_weights(const uint32_t* histo, size_t nbins, int mode, int num, int den, float* weights)
{
    for (size_t i = 0; i < nbins; ++i) weights[i] = 1.f;
    if (mode == 0) return;
    const float exponent = (float)num / den;
    size_t offset = 0;
    if (mode == 2) while (offset < nbins && !histo[offset]) ++offset;
    for (size_t i = offset; i < nbins; ++i) weights[i] = (float)pow((double)(long long)(i - offset), (double)exponent);
}

/* encoders/quantiser_utils.hpp:386-418 (setup_com):
 *   importance[i] = histo[i] * weights[i]                            (:154-168)
 *   importanceSum = (float) sum over importance accumulated in double (:400, init `0.`)
 *   n_levels <= 256 -> linear_mapping_quantisation (:286-306) else adaptive_lloyd_com (:227-284)
 * lut_encode is `char` in the reference; its bytes are what is stored, so uint8_t here. */
void sqo_quantiser_build_luts_w(const uint32_t* histo, size_t nbins, int mode, int num, int den, uint8_t* lut_encode, uint16_t* lut_decode)
{
    const size_t max_compressed = 256;
    const uint16_t raw_max = (nbins == 65536) ? 65535 : 255;
    float* importance = (float*)malloc(nbins * sizeof(float));
    float* weights = (float*)malloc(nbins * sizeof(float));
    memset(lut_encode, 0, nbins);
    memset(lut_decode, 0, max_compressed * sizeof(uint16_t));
    sqo_quantiser_weights(histo, nbins, mode, num, den, weights);
    for (size_t i = 0; i < nbins; ++i) importance[i] = (float)histo[i] * weights[i];
    free(weights);

    double acc = 0.;
    for (size_t i = 0; i < nbins; ++i) acc = acc + importance[i];
    const float importanceSum = (float)acc;
    if (!(importanceSum != 0)) { free(importance); return; }

    uint32_t n_levels = 0;
    for (size_t i = 0; i < nbins; ++i) if (importance[i] != 0.f) n_levels++;

    if (n_levels <= max_compressed) {
        uint32_t comp_idx = 0;
        for (uint32_t raw_idx = 0; raw_idx < nbins && comp_idx < max_compressed; ++raw_idx) {
            lut_encode[raw_idx] = (uint8_t)comp_idx;
            lut_decode[comp_idx] = (uint16_t)raw_idx;
            if (importance[raw_idx]) comp_idx++;
        }
        if (comp_idx < max_compressed && comp_idx > 0 && lut_decode[comp_idx] == raw_max) {
            for (size_t i = comp_idx; i < max_compressed; ++i) lut_decode[i] = lut_decode[comp_idx - 1];
        }
    } else {
        size_t levels_available = max_compressed;
        float bucketSize = importanceSum / levels_available;
        float importanceIntegral = importance[0];
        float quantile_sum = importance[0];
        uint32_t comp_idx = 0;
        float wmean = 0 * importance[0];
        float index_wmean = 0;
        for (uint32_t raw_idx = 1; raw_idx < nbins; ++raw_idx) {
            if (quantile_sum >= bucketSize && (comp_idx < max_compressed - 1)) {
                lut_decode[comp_idx] = (uint16_t)index_wmean;
                comp_idx++;
                levels_available--;
                quantile_sum = importance[raw_idx];
                wmean = raw_idx * importance[raw_idx];
                if (importanceIntegral < importanceSum)
                    bucketSize = (importanceSum - importanceIntegral) / levels_available;
                if (quantile_sum != 0.) index_wmean = roundf(wmean / quantile_sum);
            } else {
                quantile_sum += importance[raw_idx];
                wmean += raw_idx * importance[raw_idx];
                if (quantile_sum != 0.) index_wmean = roundf(wmean / quantile_sum);
            }
            lut_encode[raw_idx] = (uint8_t)comp_idx;
            importanceIntegral += importance[raw_idx];
        }
        lut_decode[comp_idx] = (uint16_t)index_wmean;
    }
    free(importance);
}

void sqo_quantiser_build_luts(const uint32_t* histo, size_t nbins, uint8_t* lut_encode, uint16_t* lut_decode)
{
    sqo_quantiser_build_luts_w(histo, nbins, 0, 1, 1, lut_encode, lut_decode);
}

/* encoders/quantiser_utils.hpp:26-42 (applyLUT) via quantiser_scheme_impl.hpp:206-223 */
void sqo_quantiser_apply_u16(const uint16_t* in, size_t len, const uint8_t* lut_encode, uint8_t* out)
{
    for (size_t i = 0; i < len; ++i) out[i] = lut_encode[in[i]];
}

void sqo_quantiser_apply_u8(const uint8_t* in, size_t len, const uint8_t* lut_encode, uint8_t* out)
{
    for (size_t i = 0; i < len; ++i) out[i] = lut_encode[in[i]];
}

/* encoders/frame_shuffle_utils.hpp:91-172 (encode_full, frame_chunk_size = 1):
 *   metric[z] = (float sequential sum of frame z) / (Y*X)      (:126-133, std::accumulate float(0))
 *   sorted = std::sort(metric)
 *   slot i <- first frame whose metric == sorted[i] (std::find) -> equal metrics map to the SAME frame
 */
static int cmp_float(const void* a, const void* b)
{
    const float fa = *(const float*)a, fb = *(const float*)b;
    return (fa > fb) - (fa < fb);
}

#define SQO_FRAME_SHUFFLE_BODY(T)                                                       \
    const size_t Z = shape[0], per = shape[1] * shape[2];                               \
    float* metric = (float*)malloc((Z ? Z : 1) * sizeof(float));                        \
    float* sorted = (float*)malloc((Z ? Z : 1) * sizeof(float));                        \
    if (!metric || !sorted) { free(metric); free(sorted); return 1; }                   \
    for (size_t z = 0; z < Z; ++z) {                                                    \
        const T* p = in + z * per;                                                      \
        float sum = 0.f;                                                                \
        for (size_t i = 0; i < per; ++i) sum = sum + (float)p[i];                       \
        metric[z] = sum / per;                                                          \
    }                                                                                   \
    memcpy(sorted, metric, Z * sizeof(float));                                          \
    qsort(sorted, Z, sizeof(float), cmp_float);                                         \
    for (size_t i = 0; i < Z; ++i) {                                                    \
        size_t src = 0;                                                                 \
        while (src < Z && !(metric[src] == sorted[i])) ++src;                           \
        decode_map[i] = (uint64_t)src;                                                  \
        memcpy(out + i * per, in + src * per, per * sizeof(T));                         \
    }                                                                                   \
    free(metric); free(sorted);                                                         \
    return 0;

int sqo_frame_shuffle_encode_u16(const uint16_t* in, uint16_t* out, const size_t shape[3], uint64_t* decode_map)
{
    SQO_FRAME_SHUFFLE_BODY(uint16_t)
}

int sqo_frame_shuffle_encode_u8(const uint8_t* in, uint8_t* out, const size_t shape[3], uint64_t* decode_map)
{
    SQO_FRAME_SHUFFLE_BODY(uint8_t)
}

/* frame_shuffle as a TAIL filter (T = char, signed on x86: the bytes of a sink's output count from -128 to 127) */
int sqo_frame_shuffle_encode_i8(const int8_t* in, int8_t* out, const size_t shape[3], uint64_t* decode_map)
{
    SQO_FRAME_SHUFFLE_BODY(int8_t)
}
