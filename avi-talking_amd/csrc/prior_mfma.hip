// Batched matrix-core DDPM sampler, 3-term bf16 split for every streamed matrix (see prior_mfma.inc).
#include "prior_mfma.inc"

// prior.hip
int avi_prior_time_table_launch(const AviPriorWeights* w, float* temb, hipStream_t s);
// prior_mfma_f16.hip
int avi_prior_sample_batched_f16_launch(const AviPriorWeights* w, const AviPriorPlanes* p, const float* text_embed,
                                        const float* noise, int B, int samples_per_group, float inv_scale, float* out,
                                        float* temb_scratch, hipStream_t s);

// prior_mfma_f16all.hip
int avi_prior_sample_batched_f16all_launch(const AviPriorWeights* w, const AviPriorPlanes* p, const float* text_embed,
                                           const float* noise, int B, int samples_per_group, float inv_scale, float* out,
                                           float* temb_scratch, hipStream_t s);

static int sample_batched_impl(const AviPriorWeights* w, const AviPriorPlanes* p, const float* text_embed,
                               const float* noise, int B, int samples_per_group, float inv_scale, float* out,
                               float* temb, bool temb_ready, void* stream) {
    if (!w || !p || !text_embed || !noise || !out || !temb || B <= 0) return AVI_EINVAL;
    if (samples_per_group < 1 || samples_per_group > SMAX) return AVI_EINVAL;
    if (w->depth < 1 || w->depth > AVI_PRIOR_MAX_DEPTH || !p->proj_hi) return AVI_EINVAL;
    // formats: a NULL lo plane = the matrix is ONE fp16 plane.  Feed-forward matrices alone (default) or every matrix.
    const bool attn16 = p->proj_lo == nullptr, ff16 = p->layer[0].w1_lo == nullptr;
    if (attn16 && !ff16) return AVI_EINVAL;
    for (int l = 0; l < w->depth; ++l) {
        const AviPriorLayerPlanes& P = p->layer[l];
        if (!P.qkv_hi || !P.out_hi || !P.w1_hi || !P.w2_hi) return AVI_EINVAL;
        if ((P.qkv_lo == nullptr) != attn16 || (P.out_lo == nullptr) != attn16) return AVI_EINVAL;
        if ((P.w1_lo == nullptr) != ff16 || (P.w2_lo == nullptr) != ff16) return AVI_EINVAL;   // same format in every layer
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (!temb_ready) {
        const int rc = avi_prior_time_table_launch(w, temb, s);
        if (rc != AVI_OK) return rc;
    }
    if (attn16)                          // opt-in: every matrix as one fp16 plane
        return avi_prior_sample_batched_f16all_launch(w, p, text_embed, noise, B, samples_per_group, inv_scale, out, temb, s);
    if (ff16)                            // feed-forward matrices as one fp16 plane
        return avi_prior_sample_batched_f16_launch(w, p, text_embed, noise, B, samples_per_group, inv_scale, out, temb, s);
    return launch_prior_variant<0>(w, p, text_embed, noise, B, samples_per_group, inv_scale, out, temb, s);
}

extern "C" int avi_prior_sample_batched(const AviPriorWeights* w, const AviPriorPlanes* p, const float* text_embed,
                                        const float* noise, int B, int samples_per_group, float inv_scale, float* out,
                                        float* temb_scratch, void* stream) {
    return sample_batched_impl(w, p, text_embed, noise, B, samples_per_group, inv_scale, out, temb_scratch, false, stream);
}

extern "C" int avi_prior_time_table(const AviPriorWeights* w, float* temb, void* stream) {
    return avi_prior_time_table_launch(w, temb, static_cast<hipStream_t>(stream));
}

extern "C" int avi_prior_sample_batched_tab(const AviPriorWeights* w, const AviPriorPlanes* p, const float* text_embed,
                                            const float* noise, int B, int samples_per_group, float inv_scale,
                                            float* out, const float* temb_table, void* stream) {
    return sample_batched_impl(w, p, text_embed, noise, B, samples_per_group, inv_scale, out,
                               const_cast<float*>(temb_table), true, stream);
}
