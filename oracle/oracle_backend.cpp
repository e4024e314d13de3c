// oracle_backend.cpp -- TEST INFRASTRUCTURE.  The C-ABI entry points the host mirror (uv-slam_amd/host) calls, answered by the
// CPU oracle instead of the HIP library, so that the SAME mirrored Estimator state machine can replay a frame sequence on the CPU
// and serve as the checker of the GPU-backed replay (tests/test_sequence_replay.py).  Built into oracle/libuvs_host_oracle.so
// together with the host sources; nothing under uv-slam_amd/ links or loads it.
#include <cstring>
#include <string>
#include "../include/uvs_solver.h"

extern "C" {
int oracle_solve(const uvs_options* opt, const uvs_window* w, int linear_mode, uvs_state* out, uvs_report* rep);
int oracle_marginalize(const uvs_options* opt, const uvs_window* w, int flag, uvs_prior* out);
int oracle_evaluate(const uvs_options* opt, const uvs_window* w, int robust, uvs_eval* out);
}

struct uvs_solver { uvs_options opt; std::string err; bool marg_pending = false; int marg_rc = 0; uvs_prior marg_out; };

extern "C" {
int uvs_abi_version(void) { return UVS_ABI_VERSION; }
void uvs_default_options(uvs_options* o) {      // config/euroc/euroc_config.yaml + the Ceres defaults of SURVEY.md Appendix B
    std::memset(o, 0, sizeof(*o));
    o->max_num_iterations = 10; o->focal_length = 461.6; o->point_sqrt_info = 461.6 / 1.6; o->line_factor = 300.0; o->vp_factor = 10.0;
    o->loss_point = 1.0; o->loss_line = 0.1; o->loss_vp = 1.0; o->gravity[2] = 9.81007;
    o->initial_trust_region_radius = 1e4; o->max_trust_region_radius = 1e16; o->min_trust_region_radius = 1e-32;
    o->min_relative_decrease = 1e-3; o->min_lm_diagonal = 1e-6; o->max_lm_diagonal = 1e32;
    o->function_tolerance = 1e-6; o->gradient_tolerance = 1e-10; o->parameter_tolerance = 1e-8;
    o->max_consecutive_invalid_steps = 5; o->jacobi_scaling = 1;
}
int uvs_create(const uvs_options* opts, int, int, int, int, int, int, uvs_solver** out) {
    if (!opts || !out) return UVS_ERR_INVALID_ARG;
    *out = new uvs_solver(); (*out)->opt = *opts;
    return UVS_OK;
}
void uvs_destroy(uvs_solver* s) { delete s; }
const char* uvs_last_error(const uvs_solver* s) { return s ? s->err.c_str() : "null solver"; }
const char* uvs_status_string(int st) { return st == UVS_OK ? "ok" : "oracle error"; }
int uvs_solve_window(uvs_solver* s, const uvs_window* w, uvs_state* out, uvs_report* rep) { return oracle_solve(&s->opt, w, 0, out, rep); }
int uvs_large_solve_fused(uvs_solver* s, const uvs_window* w, uvs_state* out, uvs_report* rep, float* loop_ms) { if (loop_ms) *loop_ms = 0.0f; return oracle_solve(&s->opt, w, 0, out, rep); }      // the host mirror's default single-window path
int uvs_evaluate(uvs_solver* s, const uvs_window* w, int robust, uvs_eval* out) { return oracle_evaluate(&s->opt, w, robust, out); }
int uvs_marginalize(uvs_solver* s, const uvs_window* w, int flag, uvs_prior* out) { return oracle_marginalize(&s->opt, w, flag, out); }
int uvs_marginalize_resident(uvs_solver* s, const uvs_window* w, int flag, uvs_prior* out) { return oracle_marginalize(&s->opt, w, flag, out); }
int uvs_marginalize_batch(uvs_solver* s, int n, const uvs_window* const* ws, const int* flags, uvs_prior* out, int* status) {      // (ABI v7: window by window here)
    int first = UVS_OK;
    for (int b = 0; b < n; ++b) { const int rc = oracle_marginalize(&s->opt, ws[b], flags[b], &out[b]); if (status) status[b] = rc; if (rc != UVS_OK && first == UVS_OK) first = rc; }
    return first;
}
// the two-halves form (since ABI v6): the oracle answers at once, the wait hands the result over
int uvs_marginalize_resident_begin(uvs_solver* s, const uvs_window* w, int flag) { if (s->marg_pending) return UVS_ERR_INVALID_ARG; s->marg_rc = oracle_marginalize(&s->opt, w, flag, &s->marg_out); s->marg_pending = true; return UVS_OK; }
int uvs_marginalize_wait(uvs_solver* s, uvs_prior* out) { if (!s->marg_pending) return UVS_ERR_INVALID_ARG; s->marg_pending = false; if (s->marg_rc == UVS_OK) *out = s->marg_out; return s->marg_rc; }
}
