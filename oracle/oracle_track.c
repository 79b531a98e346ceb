/*
 * oracle_track.c — TEST INFRASTRUCTURE (see ftk_oracle.h).  PARITY UNPINNED.
 *
 * Per-feature drivers: TrackMultipleLevel / TrackSingleLevel of the three trackers
 * (basic_klt.cpp:7-86, affine_klt.cpp:6-91, lssd_klt.cpp:7-94).
 */
#include "oracle_internal.h"

static inline int uv_outside(const float *uv, const orc_image *img) {
    return uv[0] < 0.0f || uv[0] > (float)(img->cols - 1) || uv[1] < 0.0f || uv[1] > (float)(img->rows - 1);
}

static inline uint32_t feature_limit(const orc_klt_options *opt, int32_t n) {
    return ((uint32_t)n < opt->max_track_points) ? (uint32_t)n : opt->max_track_points;
}

int orc_klt_track_pyramid(int model, const orc_klt_options *opt, const orc_image *ref_levels, const orc_image *cur_levels, int32_t n_levels,
                          const float *ref_uv, float *cur_uv, uint8_t *status, int32_t n, const float *prior, int consider_luminance,
                          uint32_t *iters) {
    if (n <= 0 || n_levels <= 0) {
        return 0; /* optical_flow.cpp:8-9 */
    }
    const uint32_t max_feature_id = feature_limit(opt, n);
    const float scale = (float)(1 << (n_levels - 1));
    const int fast = (opt->method != ORC_INVERSE && opt->method != ORC_DIRECT);

    for (uint32_t id = 0; id < max_feature_id; ++id) {
        uint32_t it = 0;
        if (iters) {
            iters[id] = 0;
        }
        if (status[id] > ORC_TRACKED) {
            continue; /* basic_klt.cpp:15 */
        }
        float sref[2] = {ref_uv[2 * id] / scale, ref_uv[2 * id + 1] / scale};
        float scur[2] = {cur_uv[2 * id] / scale, cur_uv[2 * id + 1] / scale};
        float affine[4] = {1.0f, 0.0f, 0.0f, 1.0f}; /* affine_klt.cpp:21 — prediction ignored in the pyramid path */
        float r[4] = {prior[0], prior[1], prior[2], prior[3]};
        float t[2];
        /* lssd_klt.cpp:22-23: t_cr = scaled_cur - predict_R_cr * scaled_ref */
        t[0] = scur[0] - (prior[0] * sref[0] + prior[1] * sref[1]);
        t[1] = scur[1] - (prior[2] * sref[0] + prior[3] * sref[1]);

        for (int32_t level = n_levels - 1; level > -1; --level) {
            const orc_image *ref = &ref_levels[level];
            const orc_image *cur = &cur_levels[level];
            switch (model) {
                case ORC_BASIC:
                    if (fast) {
                        orc_basic_track_one_fast(opt, ref, cur, sref[0], sref[1], scur, &status[id], &it);
                    } else {
                        orc_basic_track_one(opt, ref, cur, sref[0], sref[1], scur, &status[id], &it);
                    }
                    break;
                case ORC_AFFINE:
                    if (fast) {
                        orc_affine_track_one_fast(opt, ref, cur, sref[0], sref[1], scur, affine, &status[id], &it);
                    } else {
                        orc_affine_track_one(opt, ref, cur, sref[0], sref[1], scur, affine, &status[id], &it);
                    }
                    break;
                default:
                    if (fast) {
                        orc_lssd_track_one_fast(opt, ref, cur, sref[0], sref[1], r, t, consider_luminance, &status[id], &it);
                    } else {
                        orc_lssd_track_one(opt, ref, cur, sref[0], sref[1], r, t, &status[id], &it);
                    }
                    break;
            }

            if (level == 0) {
                if (model == ORC_LSSD) {
                    /* lssd_klt.cpp:43: written back with the UNSCALED ref */
                    cur_uv[2 * id] = (r[0] * ref_uv[2 * id] + r[1] * ref_uv[2 * id + 1]) + t[0];
                    cur_uv[2 * id + 1] = (r[2] * ref_uv[2 * id] + r[3] * ref_uv[2 * id + 1]) + t[1];
                } else {
                    cur_uv[2 * id] = scur[0];
                    cur_uv[2 * id + 1] = scur[1];
                }
                break;
            }
            sref[0] *= 2.0f;
            sref[1] *= 2.0f;
            if (model == ORC_LSSD) {
                t[0] *= 2.0f;
                t[1] *= 2.0f;
            } else {
                scur[0] *= 2.0f;
                scur[1] *= 2.0f;
            }
        }

        if (uv_outside(&cur_uv[2 * id], &cur_levels[0])) {
            status[id] = ORC_OUTSIDE; /* basic_klt.cpp:49-53 */
        }
        if (iters) {
            iters[id] = it;
        }
    }
    return 1;
}

int orc_klt_track_single(int model, const orc_klt_options *opt, const orc_image *ref, const orc_image *cur, const float *ref_uv, float *cur_uv,
                         uint8_t *status, int32_t n, const float *prior, int consider_luminance, uint32_t *iters) {
    if (n <= 0) {
        return 0; /* optical_flow.cpp:30 */
    }
    const uint32_t max_feature_id = feature_limit(opt, n);
    const int fast = (opt->method != ORC_INVERSE && opt->method != ORC_DIRECT);

    for (uint32_t id = 0; id < max_feature_id; ++id) {
        uint32_t it = 0;
        if (iters) {
            iters[id] = 0;
        }
        if (status[id] > ORC_TRACKED) {
            continue;
        }
        const float ru = ref_uv[2 * id], rv = ref_uv[2 * id + 1];
        switch (model) {
            case ORC_BASIC:
                if (fast) {
                    orc_basic_track_one_fast(opt, ref, cur, ru, rv, &cur_uv[2 * id], &status[id], &it);
                } else {
                    orc_basic_track_one(opt, ref, cur, ru, rv, &cur_uv[2 * id], &status[id], &it);
                }
                break;
            case ORC_AFFINE: {
                float affine[4] = {prior[0], prior[1], prior[2], prior[3]}; /* affine_klt.cpp:70 */
                if (fast) {
                    orc_affine_track_one_fast(opt, ref, cur, ru, rv, &cur_uv[2 * id], affine, &status[id], &it);
                } else {
                    orc_affine_track_one(opt, ref, cur, ru, rv, &cur_uv[2 * id], affine, &status[id], &it);
                }
                break;
            }
            default: {
                /* lssd_klt.cpp:72-89: the single-level path never writes cur_pixel_uv back (sic) */
                float r[4] = {prior[0], prior[1], prior[2], prior[3]};
                float t[2];
                t[0] = cur_uv[2 * id] - (prior[0] * ru + prior[1] * rv);
                t[1] = cur_uv[2 * id + 1] - (prior[2] * ru + prior[3] * rv);
                if (fast) {
                    orc_lssd_track_one_fast(opt, ref, cur, ru, rv, r, t, consider_luminance, &status[id], &it);
                } else {
                    orc_lssd_track_one(opt, ref, cur, ru, rv, r, t, &status[id], &it);
                }
                break;
            }
        }
        if (uv_outside(&cur_uv[2 * id], cur)) {
            status[id] = ORC_OUTSIDE;
        }
        if (iters) {
            iters[id] = it;
        }
    }
    return 1;
}
