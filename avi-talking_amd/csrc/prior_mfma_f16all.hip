// Batched matrix-core DDPM sampler with EVERY streamed matrix stored as one fp16 plane (see prior_mfma.inc): the opt-in
// AVI_PRIOR_ATTN_FP16=1 variant, 30 % fewer bytes per DDPM step than the default (feed-forward matrices only).  Its own
// translation unit = its own code object, like the other two variants (prior_mfma.inc header).
#include "prior_mfma.inc"

int avi_prior_sample_batched_f16all_launch(const AviPriorWeights* w, const AviPriorPlanes* p, const float* text_embed,
                                           const float* noise, int B, int samples_per_group, float inv_scale, float* out,
                                           float* temb_scratch, hipStream_t s) {
    return launch_prior_variant<2>(w, p, text_embed, noise, B, samples_per_group, inv_scale, out, temb_scratch, s);
}
