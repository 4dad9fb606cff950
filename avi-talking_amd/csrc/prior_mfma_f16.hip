// Batched matrix-core DDPM sampler with the feed-forward matrices stored as ONE fp16 plane (see prior_mfma.inc).
#include "prior_mfma.inc"

int avi_prior_sample_batched_f16_launch(const AviPriorWeights* w, const AviPriorPlanes* p, const float* text_embed,
                                        const float* noise, int B, int samples_per_group, float inv_scale, float* out,
                                        float* temb_scratch, hipStream_t s) {
    return launch_prior_variant<1>(w, p, text_embed, noise, B, samples_per_group, inv_scale, out, temb_scratch, s);
}
