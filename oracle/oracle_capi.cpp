// ORACLE — TEST INFRASTRUCTURE ONLY.
// extern "C" surface of the CPU restatement (liboracle.so), loaded through ctypes by tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg — never by the product path.
// All pointers are host pointers; matrices are column-major (Eigen .data() order).
#include <omp.h>

#include <random>

#include "oracle_features.hpp"
#include "oracle_knn.hpp"
#include "oracle_math.hpp"
#include "oracle_registration.hpp"
#include "oracle_robust_cov.hpp"
#include "oracle_voxel_hash_map.hpp"

using namespace oracle;

extern "C" {

int orc_num_threads() { return omp_get_max_threads(); }
void orc_set_num_threads(int n) { omp_set_num_threads(n); }

// ---- synthetic clouds: the reference tests' idiom (cpp/tests/test_kdtree.cpp:69-75), x then y then z.
// `state` lets a caller continue the same generator across calls (targets, then queries).
void* orc_rng_new(uint32_t seed) { return new std::mt19937(seed); }
void orc_rng_free(void* g) { delete static_cast<std::mt19937*>(g); }
void orc_rng_uniform_points(void* g, float range, size_t n, float* out) {
    auto& gen = *static_cast<std::mt19937*>(g);
    std::uniform_real_distribution<float> dist(-range, range);
    for (size_t i = 0; i < n; ++i) {
        const float x = dist(gen);
        const float y = dist(gen);
        const float z = dist(gen);
        out[4 * i + 0] = x; out[4 * i + 1] = y; out[4 * i + 2] = z; out[4 * i + 3] = 1.0f;
    }
}
void orc_rng_normal(void* g, float stddev, size_t n, float* out) {
    auto& gen = *static_cast<std::mt19937*>(g);
    std::normal_distribution<float> dist(0.0f, stddev);
    for (size_t i = 0; i < n; ++i) out[i] = dist(gen);
}
// random_sampling_operator.hpp:36-42 — partial Fisher-Yates with std::mt19937 + uniform_int_distribution<size_t>
void orc_random_sampling_flags(uint32_t seed, size_t N, size_t num, uint8_t* flags) {
    std::mt19937 mt(seed);
    std::vector<size_t> indices(N);
    std::iota(indices.begin(), indices.end(), 0);
    for (size_t i = 0; i < N; ++i) flags[i] = 0;
    if (N <= num) { for (size_t i = 0; i < N; ++i) flags[i] = 1; return; }
    for (size_t i = 0; i < num; ++i) {
        std::uniform_int_distribution<size_t> dist(i, N - 1);
        const size_t j = dist(mt);
        std::swap(indices[i], indices[j]);
    }
    for (size_t i = 0; i < num; ++i) flags[indices[i]] = 1;
}

// ---- math probes (for pinning against cpp/tests/test_eigen_utils.cpp)
void orc_eigen3(const float* A_colmajor9, float* vals3, float* vecs_colmajor9) {
    Mat3 A; std::memcpy(A.d, A_colmajor9, 36);
    Vec3 v; Mat3 V;
    symmetric_eigen_decomposition_3x3(A, v, V);
    std::memcpy(vals3, v.d, 12); std::memcpy(vecs_colmajor9, V.d, 36);
}
void orc_inverse3(const float* A, float* out) { Mat3 m; std::memcpy(m.d, A, 36); const Mat3 r = inverse(m); std::memcpy(out, r.d, 36); }
float orc_det3(const float* A) { Mat3 m; std::memcpy(m.d, A, 36); return determinant(m); }
// Small eigen_utils helpers used inside K11 (test_eigen_utils.cpp:471-595): op 0 dot<3>, 1 dot<4>, 2 cross, 3 outer<4>,
// 4 transpose<3,3>, 5 transpose<4,6>, 6 ensure_symmetric<3>, 7 frobenius_norm<3,3>, 8 frobenius_norm<3> (vector),
// 9 frobenius_norm_squared<3> (vector), 10 element_wise_multiply<3,3>, 11 element_wise_multiply<4,4>.
// Operands and results are column-major, as everywhere in this file.
void orc_eigen_util(int op, const float* a, const float* b, float* out) {
    auto ld3 = [](const float* p) { Vec3 v; std::memcpy(v.d, p, 12); return v; };
    auto ld4 = [](const float* p) { Vec4 v; std::memcpy(v.d, p, 16); return v; };
    auto ldm3 = [](const float* p) { Mat3 m; std::memcpy(m.d, p, 36); return m; };
    auto ldm4 = [](const float* p) { Mat4 m; std::memcpy(m.d, p, 64); return m; };
    switch (op) {
        case 0: out[0] = dot<3>(ld3(a), ld3(b)); break;
        case 1: out[0] = dot<4>(ld4(a), ld4(b)); break;
        case 2: { const Vec3 r = cross(ld3(a), ld3(b)); std::memcpy(out, r.d, 12); break; }
        case 3: { const Mat4 r = outer<4>(ld4(a), ld4(b)); std::memcpy(out, r.d, 64); break; }
        case 4: { const Mat3 r = transpose<3, 3>(ldm3(a)); std::memcpy(out, r.d, 36); break; }
        case 5: { Mat<4, 6> m; std::memcpy(m.d, a, 96); const Mat<6, 4> r = transpose<4, 6>(m); std::memcpy(out, r.d, 96); break; }
        case 6: { const Mat3 r = ensure_symmetric<3>(ldm3(a)); std::memcpy(out, r.d, 36); break; }
        case 7: out[0] = frobenius_norm<3, 3>(ldm3(a)); break;
        case 8: out[0] = norm<3>(ld3(a)); break;
        case 9: out[0] = norm_squared<3>(ld3(a)); break;
        case 10: { const Mat3 r = element_wise_multiply<3, 3>(ldm3(a), ldm3(b)); std::memcpy(out, r.d, 36); break; }
        case 11: { const Mat4 r = element_wise_multiply<4, 4>(ldm4(a), ldm4(b)); std::memcpy(out, r.d, 64); break; }
        default: break;
    }
}
void orc_matmul4(const float* A, const float* B, float* out) {
    Mat4 a, b; std::memcpy(a.d, A, 64); std::memcpy(b.d, B, 64);
    const Mat4 r = matmul<4, 4, 4>(a, b); std::memcpy(out, r.d, 64);
}
// the annealing wrapper's schedule (oracle_registration.hpp: robust_annealing_scales); returns the number of levels written
int orc_robust_annealing_scales(int loss_is_none, int auto_scale, float default_scale, float init_scale, float min_scale,
                                int auto_scaling_iter, float* scales_out, int capacity) {
    const std::vector<float> s = robust_annealing_scales(loss_is_none != 0, auto_scale != 0, default_scale, init_scale, min_scale,
                                                         (size_t)std::max(0, auto_scaling_iter));
    for (int i = 0; i < (int)s.size() && i < capacity; ++i) scales_out[i] = s[i];
    return (int)s.size();
}
void orc_se3_exp(const float* twist6, float* T16) { Vec6 a; std::memcpy(a.d, twist6, 24); const Mat4 T = se3_exp(a); std::memcpy(T16, T.d, 64); }
void orc_se3_log(const float* T16, float* twist6) { Mat4 T; std::memcpy(T.d, T16, 64); const Vec6 a = se3_log(T); std::memcpy(twist6, a.d, 24); }
void orc_so3_exp(const float* w3, float* q4) { Vec3 w; std::memcpy(w.d, w3, 12); const Vec4 q = so3_exp(w); std::memcpy(q4, q.d, 16); }
void orc_so3_log(const float* q4, float* w3) { Vec4 q; std::memcpy(q.d, q4, 16); const Vec3 w = so3_log(q); std::memcpy(w3, w.d, 12); }
void orc_isometry_mul(const float* A, const float* B, float* out) {
    const Mat4 r = isometry_mul(to_mat4(A), to_mat4(B)); std::memcpy(out, r.d, 64);
}
int orc_ldlt6_solve(const float* H36_colmajor, const float* b6, float* x6) {
    Mat6 H; Vec6 b, x; std::memcpy(H.d, H36_colmajor, 144); std::memcpy(b.d, b6, 24);
    const bool ok = ldlt6_solve(H, b, x); std::memcpy(x6, x.d, 24); return ok ? 1 : 0;
}
float orc_robust_weight(int loss, float r, float s) { return robust_weight(loss, r, s); }
float orc_robust_error(int loss, float r, float s) { return robust_error(loss, r, s); }

// ---- KNN
void orc_knn_bruteforce(const float* q, size_t nq, const float* t, size_t nt, size_t k, int32_t* idx, float* d2) {
    knn_bruteforce(q, nq, t, nt, k, idx, d2);
}
// nodes_out must hold 2*n nodes of 32 bytes; returns the used node count.
size_t orc_kdtree_build(const float* pts, size_t n, size_t leaf_threshold, void* nodes_out) {
    const std::vector<FlatKDNode> tree = kdtree_build(pts, n, leaf_threshold);
    if (!tree.empty()) std::memcpy(nodes_out, tree.data(), tree.size() * sizeof(FlatKDNode));
    return tree.size();
}
int orc_kdtree_knn(const void* nodes, size_t n_nodes, const float* q, size_t nq, size_t k, const float* T16, int32_t* idx,
                   float* d2) {
    if (kdtree_max_k_class(k) == 0) return 1;
    kdtree_search(static_cast<const FlatKDNode*>(nodes), n_nodes, q, nq, k, T16, idx, d2, -1.0f);
    return 0;
}
int orc_kdtree_radius(const void* nodes, size_t n_nodes, const float* q, size_t nq, size_t max_k, float radius,
                      const float* T16, int32_t* idx, float* d2) {
    if (kdtree_max_k_class(max_k) == 0) return 1;
    kdtree_search(static_cast<const FlatKDNode*>(nodes), n_nodes, q, nq, max_k, T16, idx, d2, radius * radius);
    return 0;
}
void orc_kdtree_remove_by_flags(void* nodes, size_t n_nodes, const uint8_t* flags, const int32_t* new_idx, size_t nflags) {
    kdtree_remove_by_flags(static_cast<FlatKDNode*>(nodes), n_nodes, flags, new_idx, nflags);
}

// ---- features
void orc_cov_estimate(const float* pts, size_t n, const int32_t* idx, size_t k, float* covs) {
#pragma omp parallel for schedule(static)
    for (long long i = 0; i < (long long)n; ++i) cov_estimate_one(covs + 16 * i, pts, k, idx, (size_t)i);
}
void orc_cov_estimate_robust(const float* pts, size_t n, const int32_t* idx, size_t k, int robust_type, float mad_scale,
                             float min_robust_scale, size_t max_iter, float* covs) {
#pragma omp parallel for schedule(static)
    for (long long i = 0; i < (long long)n; ++i)
        cov_estimate_robust_one(covs + 16 * i, pts, k, idx, (size_t)i, robust_type, mad_scale, min_robust_scale, max_iter);
}
void orc_cov_normalize(float* covs, size_t n) {
#pragma omp parallel for schedule(static)
    for (long long i = 0; i < (long long)n; ++i) normalize_covariance(covs + 16 * i);
}
void orc_normals_from_knn(const float* pts, size_t n, const int32_t* idx, size_t k, float* normals) {
#pragma omp parallel for schedule(static)
    for (long long i = 0; i < (long long)n; ++i) {
        float cov[16];
        cov_estimate_one(cov, pts, k, idx, (size_t)i);
        extract_normal(pts + 4 * i, cov, normals + 4 * i);
    }
}
void orc_normals_from_cov(const float* pts, const float* covs, size_t n, float* normals) {
#pragma omp parallel for schedule(static)
    for (long long i = 0; i < (long long)n; ++i) extract_normal(pts + 4 * i, covs + 16 * i, normals + 4 * i);
}
void orc_update_covariance_plane(float* covs, size_t n) {
#pragma omp parallel for schedule(static)
    for (long long i = 0; i < (long long)n; ++i) update_covariance_plane(covs + 16 * i);
}
void orc_normalize_covariance(float* covs, size_t n) {
    for (size_t i = 0; i < n; ++i) normalize_covariance(covs + 16 * i);
}
void orc_transform_points(const float* in, float* out, size_t n, const float* T16) {
    for (size_t i = 0; i < n; ++i) { float r[4]; transform_point(in + 4 * i, r, T16); std::memcpy(out + 4 * i, r, 16); }
}
void orc_transform_covs(const float* in, float* out, size_t n, const float* T16) {
    for (size_t i = 0; i < n; ++i) { float r[16]; transform_cov(in + 16 * i, r, T16); std::memcpy(out + 16 * i, r, 64); }
}
void orc_transform_normals(const float* in, float* out, size_t n, const float* T16) {
    for (size_t i = 0; i < n; ++i) { float r[4]; transform_normal(in + 4 * i, r, T16); std::memcpy(out + 4 * i, r, 16); }
}
void orc_voxel_keys(const float* pts, size_t n, float voxel_size, uint64_t* keys) {
    const float inv = 1.0f / voxel_size;
#pragma omp parallel for schedule(static)
    for (long long i = 0; i < (long long)n; ++i) keys[i] = compute_voxel_bit(pts + 4 * i, inv);
}
// Outputs must hold n entries (worst case); optional attributes may be null. Returns the voxel count.
size_t orc_voxel_downsample(const float* pts, size_t n, float voxel_size, size_t min_count, const float* rgb,
                            const float* intensity, const float* ts, int stable, float* out_pts, float* out_rgb,
                            float* out_intensity, float* out_ts, uint64_t* out_keys) {
    const VoxelOut o = voxel_downsample(pts, n, voxel_size, min_count, rgb, intensity, ts, stable != 0);
    const size_t v = o.keys.size();
    if (v) {
        std::memcpy(out_pts, o.points.data(), v * 16);
        if (rgb && out_rgb) std::memcpy(out_rgb, o.rgb.data(), v * 16);
        if (intensity && out_intensity) std::memcpy(out_intensity, o.intensities.data(), v * 4);
        if (ts && out_ts) std::memcpy(out_ts, o.timestamps.data(), v * 4);
        if (out_keys) std::memcpy(out_keys, o.keys.data(), v * 8);
    }
    return v;
}
void orc_box_filter(const float* pts, size_t n, float min_d, float max_d, uint8_t* flags) {
    for (size_t i = 0; i < n; ++i) flags[i] = box_filter_flag(pts + 4 * i, min_d, max_d);
}

// ---- registration
static FactorParams make_fp(int reg, int loss, float max_corr, float genz_thr) {
    FactorParams fp;
    fp.reg_type = reg; fp.robust_type = loss; fp.max_correspondence_distance = max_corr; fp.genz_planarity_threshold = genz_thr;
    return fp;
}
// out44: H row-major [0..35], b [36..41], error [42], inlier as uint32 bits [43]. per_point may be null (n*44 floats).
void orc_gicp_linearize(const float* src, const float* src_cov, size_t n, const float* tgt, const float* tgt_cov,
                        const float* tgt_nrm, const int32_t* nn_idx, const float* nn_d2, const float* T16, float max_corr,
                        int reg, int loss, float robust_scale, float genz_alpha, float* out44, float* per_point) {
    Cloud s, t;
    s.points = src; s.covs = src_cov; s.n = n;
    t.points = tgt; t.covs = tgt_cov; t.normals = tgt_nrm;
    const Linearized L = linearize_reduce(make_fp(reg, loss, max_corr, 0.2f), s, t, nn_idx, nn_d2, T16, robust_scale, genz_alpha, per_point);
    for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) out44[r * 6 + c] = L.H(r, c);
    for (int r = 0; r < 6; ++r) out44[36 + r] = L.b[r];
    out44[42] = L.error;
    std::memcpy(out44 + 43, &L.inlier, 4);
}
void orc_gicp_error(const float* src, const float* src_cov, size_t n, const float* tgt, const float* tgt_cov,
                    const float* tgt_nrm, const int32_t* nn_idx, const float* nn_d2, const float* T16, float max_corr, int reg,
                    int loss, float robust_scale, float genz_alpha, float* out2) {
    Cloud s, t;
    s.points = src; s.covs = src_cov; s.n = n;
    t.points = tgt; t.covs = tgt_cov; t.normals = tgt_nrm;
    float e; uint32_t inl;
    error_reduce(make_fp(reg, loss, max_corr, 0.2f), s, t, nn_idx, nn_d2, T16, robust_scale, genz_alpha, e, inl);
    out2[0] = e; std::memcpy(out2 + 1, &inl, 4);
}
// K11 / K12 with the rotation-constraint term (registration.hpp:630-650, 758-766)
void orc_gicp_linearize_rot(const float* src, const float* src_cov, size_t n, const float* tgt, const float* tgt_cov,
                            const float* tgt_nrm, const int32_t* nn_idx, const float* nn_d2, const float* T16, float max_corr,
                            int reg, int loss, float robust_scale, float genz_alpha, float rot_weight, float rot_scale,
                            float* out44) {
    Cloud s, t;
    s.points = src; s.covs = src_cov; s.n = n;
    t.points = tgt; t.covs = tgt_cov; t.normals = tgt_nrm;
    FactorParams fp = make_fp(reg, loss, max_corr, 0.2f);
    fp.rot_enable = true; fp.rot_weight = rot_weight;
    const Linearized L = linearize_reduce(fp, s, t, nn_idx, nn_d2, T16, robust_scale, genz_alpha, nullptr, rot_scale);
    for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) out44[r * 6 + c] = L.H(r, c);
    for (int r = 0; r < 6; ++r) out44[36 + r] = L.b[r];
    out44[42] = L.error;
    std::memcpy(out44 + 43, &L.inlier, 4);
}
void orc_gicp_error_rot(const float* src, const float* src_cov, size_t n, const float* tgt, const float* tgt_cov,
                        const float* tgt_nrm, const int32_t* nn_idx, const float* nn_d2, const float* T16, float max_corr,
                        int reg, int loss, float robust_scale, float genz_alpha, float rot_weight, float rot_scale,
                        float* out2) {
    Cloud s, t;
    s.points = src; s.covs = src_cov; s.n = n;
    t.points = tgt; t.covs = tgt_cov; t.normals = tgt_nrm;
    FactorParams fp = make_fp(reg, loss, max_corr, 0.2f);
    fp.rot_enable = true; fp.rot_weight = rot_weight;
    float e; uint32_t inl;
    error_reduce(fp, s, t, nn_idx, nn_d2, T16, robust_scale, genz_alpha, e, inl, rot_scale);
    out2[0] = e; std::memcpy(out2 + 1, &inl, 4);
}
// degenerate_regularization.hpp:60-110. H36 column-major (symmetric anyway), in place.
void orc_degenerate_regularize(int type, float rot_thr, float trans_thr, float base_factor, float* H36, float* b6,
                               uint32_t inlier, const float* T_cur16, const float* T_init16) {
    DegenerateRegParams p;
    p.type = type; p.rot_eigenvalue_threshold = rot_thr; p.trans_eigenvalue_threshold = trans_thr; p.base_factor = base_factor;
    Mat6 H; Vec6 b;
    std::memcpy(H.d, H36, 144); std::memcpy(b.d, b6, 24);
    degenerate_regularize(p, H, b, inlier, to_mat4(T_cur16), to_mat4(T_init16));
    std::memcpy(H36, H.d, 144); std::memcpy(b6, b.d, 24);
}
// map_prior.hpp:97-174. sig4 = rot_vel, trans_vel, rot_base, trans_base. Returns has_prior.
int orc_map_prior_update(const float* sig4, const float* H_raw36, float error_raw, uint32_t inlier, const float* T_prev16,
                         const float* T_pred16, float* omega36, float* T_pred_inv16) {
    MapPrior mp;
    mp.params.enabled = true;
    mp.params.rot_vel_sigma = sig4[0]; mp.params.trans_vel_sigma = sig4[1];
    mp.params.rot_base_sigma = sig4[2]; mp.params.trans_base_sigma = sig4[3];
    Mat6 H; std::memcpy(H.d, H_raw36, 144);
    mp.update(H, error_raw, inlier, to_mat4(T_prev16), to_mat4(T_pred16));
    std::memcpy(omega36, mp.Omega.d, 144);
    std::memcpy(T_pred_inv16, mp.T_pred_inv.d, 64);
    return mp.has_prior ? 1 : 0;
}
// map_prior.hpp:181-201. H/b/error updated in place; returns prior_error(T_est).
float orc_map_prior_apply(const float* omega36, const float* T_pred_inv16, float* H36, float* b6, float* error,
                          const float* T_est16) {
    MapPrior mp;
    mp.params.enabled = true; mp.has_prior = true;
    std::memcpy(mp.Omega.d, omega36, 144); std::memcpy(mp.T_pred_inv.d, T_pred_inv16, 64);
    if (H36) {
        Mat6 H; Vec6 b;
        std::memcpy(H.d, H36, 144); std::memcpy(b.d, b6, 24);
        mp.apply(H, b, *error, to_mat4(T_est16));
        std::memcpy(H36, H.d, 144); std::memcpy(b6, b.d, 24);
    }
    return mp.prior_error(to_mat4(T_est16));
}
void orc_icp_robust_weights(const float* src, const float* src_cov, size_t n, const float* tgt, const float* tgt_cov,
                            const float* tgt_nrm, const int32_t* nn_idx, const float* nn_d2, const float* T16, float max_corr,
                            int reg, int loss, float robust_scale, float* out) {
    Cloud s, t;
    s.points = src; s.covs = src_cov; s.n = n;
    t.points = tgt; t.covs = tgt_cov; t.normals = tgt_nrm;
    icp_robust_weights(make_fp(reg, loss, max_corr, 0.2f), s, t, nn_idx, nn_d2, T16, robust_scale, 1.0f, out);
}

struct orc_reg_params {
    int reg_type, robust_type, optimization_method, max_iterations;
    float max_correspondence_distance, robust_default_scale, gn_lambda;
    float lm_init_lambda, lm_lambda_factor, lm_min_lambda, lm_max_lambda;
    int lm_max_inner_iterations;
    float crit_translation, crit_rotation;
    // annealing wrapper (pipeline/robust.hpp); auto_scale=0 -> plain align with robust_default_scale
    int auto_scale, auto_scaling_iter;
    float init_scale, min_scale;
    // Powell dogleg (registration_params.hpp:84-92); dl_initial_radius == 0 keeps the reference defaults
    float dl_initial_radius, dl_min_radius, dl_max_radius, dl_eta1, dl_eta2, dl_gamma_decrease, dl_gamma_increase;
    // default-off terms (all zero = off): rotation constraint, degenerate regularisation, MAP prior
    int rot_enable;
    float rot_weight, rot_robust_default_scale;
    int dr_type;
    float dr_rot_threshold, dr_trans_threshold, dr_base_factor;
    int mp_active;          // a prior computed by orc_map_prior_update
    float mp_omega[36];     // column-major
    float mp_T_pred_inv[16];
};
struct orc_reg_result {
    float T[16];
    float H[36];  // column-major
    float b[6];
    float error;
    uint32_t inlier;
    int iterations;
    int converged;
    float H_raw[36];  // column-major
    float b_raw[6];
    float error_raw;
};
// nn_mode: 0 = KD-tree built on target (the reference's default KNNBase), 1 = brute force.
// trace_T (optional, max_iterations*16 floats) receives the pose after every outer iteration; *trace_n their count.
// prebuilt_nodes (optional): a KD-tree from orc_kdtree_build on the same target, so the build is not part of the call.
void orc_registration_align(const orc_reg_params* P, const float* src, const float* src_cov, size_t ns, const float* tgt,
                            const float* tgt_cov, const float* tgt_nrm, size_t nt, const float* init_T16, int nn_mode,
                            orc_reg_result* out, float* trace_T, int* trace_n, const void* prebuilt_nodes,
                            size_t prebuilt_n_nodes, float* trace_steps, int trace_steps_capacity, int* trace_steps_n) {
    // trace_steps (optional, capacity entries of 5 floats): per outer iteration — all annealing levels in order — {trial
    // evaluations, accepted, lambda / trust radius after the iteration, result.error after it, decision margin}; *trace_steps_n their count
    RegParams p;
    p.reg_type = P->reg_type; p.robust_type = P->robust_type; p.optimization_method = P->optimization_method;
    p.max_iterations = (size_t)P->max_iterations; p.max_correspondence_distance = P->max_correspondence_distance;
    p.robust_default_scale = P->robust_default_scale; p.gn_lambda = P->gn_lambda;
    p.lm_init_lambda = P->lm_init_lambda; p.lm_lambda_factor = P->lm_lambda_factor;
    p.lm_min_lambda = P->lm_min_lambda; p.lm_max_lambda = P->lm_max_lambda;
    p.lm_max_inner_iterations = (size_t)P->lm_max_inner_iterations;
    p.crit_translation = P->crit_translation; p.crit_rotation = P->crit_rotation;
    if (P->dl_initial_radius > 0.0f) {
        p.dl_initial_radius = P->dl_initial_radius; p.dl_min_radius = P->dl_min_radius; p.dl_max_radius = P->dl_max_radius;
        p.dl_eta1 = P->dl_eta1; p.dl_eta2 = P->dl_eta2; p.dl_gamma_decrease = P->dl_gamma_decrease;
        p.dl_gamma_increase = P->dl_gamma_increase;
    }
    p.rot_enable = P->rot_enable != 0;
    if (p.rot_enable) { p.rot_weight = P->rot_weight; p.rot_robust_default_scale = P->rot_robust_default_scale; }
    p.degenerate_reg.type = P->dr_type;
    if (P->dr_type) {
        p.degenerate_reg.rot_eigenvalue_threshold = P->dr_rot_threshold;
        p.degenerate_reg.trans_eigenvalue_threshold = P->dr_trans_threshold;
        p.degenerate_reg.base_factor = P->dr_base_factor;
    }
    MapPrior prior;
    if (P->mp_active) {
        prior.params.enabled = true;
        prior.has_prior = true;
        std::memcpy(prior.Omega.d, P->mp_omega, 144);
        std::memcpy(prior.T_pred_inv.d, P->mp_T_pred_inv, 64);
    }
    Cloud s, t;
    s.points = src; s.covs = src_cov; s.n = ns;
    t.points = tgt; t.covs = tgt_cov; t.normals = tgt_nrm; t.n = nt;
    std::vector<FlatKDNode> tree;
    const FlatKDNode* nodes = static_cast<const FlatKDNode*>(prebuilt_nodes);
    size_t n_nodes = prebuilt_n_nodes;
    if (nn_mode == 0 && nodes == nullptr) {  // KDTree::build is target preprocessing in the reference's flow
        tree = kdtree_build(tgt, nt, 16);
        nodes = tree.data();
        n_nodes = tree.size();
    }
    NearestFn nearest = [&](const float* q, size_t nq, const float* T, int32_t* idx, float* d2) {
        if (nn_mode == 0) {
            kdtree_search(nodes, n_nodes, q, nq, 1, T, idx, d2, -1.0f);
        } else {
            std::vector<float> tq(nq * 4);
            for (size_t i = 0; i < nq; ++i) transform_point(q + 4 * i, tq.data() + 4 * i, T);
            knn_bruteforce(tq.data(), nq, tgt, nt, 1, idx, d2);
        }
    };
    std::vector<float> trace, steps;
    RegResult r;
    if (P->auto_scale)
        r = align_robust_annealing(p, s, t, nearest, init_T16, true, P->init_scale, P->min_scale, (size_t)P->auto_scaling_iter,
                                   trace_steps ? &steps : nullptr);
    else
        r = align(p, s, t, nearest, init_T16, -1.0f, trace_T ? &trace : nullptr, P->mp_active ? &prior : nullptr, -1.0f,
                  trace_steps ? &steps : nullptr);
    if (trace_steps) {
        const size_t k = std::min(steps.size() / 5, (size_t)std::max(trace_steps_capacity, 0));
        std::memcpy(trace_steps, steps.data(), k * 20);
        if (trace_steps_n) *trace_steps_n = (int)k;
    }
    std::memcpy(out->H_raw, r.H_raw.d, 144);
    std::memcpy(out->b_raw, r.b_raw.d, 24);
    out->error_raw = r.error_raw;
    std::memcpy(out->T, r.T.d, 64);
    std::memcpy(out->H, r.H.d, 144);
    std::memcpy(out->b, r.b.d, 24);
    out->error = r.error; out->inlier = r.inlier; out->iterations = (int)r.iterations; out->converged = r.converged ? 1 : 0;
    if (trace_T) { std::memcpy(trace_T, trace.data(), trace.size() * 4); *trace_n = (int)(trace.size() / 16); }
}


// ---- VoxelHashMap (algorithms/mapping/voxel_hash_map.hpp). param: 0 voxel_size, 1 max_staleness, 2 remove_old_data_cycle,
// 3 rehash_threshold, 4 min_num_point. info: 0 voxel_num, 1 capacity, 2 staleness_counter, 3 has_cov, 4 has_rgb, 5 has_intensity.
void* orc_vhm_new(float voxel_size) { return voxel_size > 0.0f ? new VoxelHashMap(voxel_size) : nullptr; }
void orc_vhm_free(void* m) { delete static_cast<VoxelHashMap*>(m); }
void orc_vhm_clear(void* m) { static_cast<VoxelHashMap*>(m)->clear(); }
void orc_vhm_set(void* m, int param, float v) {
    auto& M = *static_cast<VoxelHashMap*>(m);
    if (param == 0) { M.voxel_size = v; M.voxel_size_inv = 1.0f / v; }
    else if (param == 1) M.max_staleness = (uint32_t)v;
    else if (param == 2) M.remove_old_data_cycle = (uint32_t)v;
    else if (param == 3) M.rehash_threshold = v;
    else if (param == 4) M.min_num_point = (uint32_t)v;
}
size_t orc_vhm_info(void* m, int what) {
    auto& M = *static_cast<VoxelHashMap*>(m);
    switch (what) {
        case 0: return M.voxel_num;
        case 1: return M.capacity;
        case 2: return M.staleness_counter;
        case 3: return M.has_cov;
        case 4: return M.has_rgb;
        case 5: return M.has_intensity;
    }
    return 0;
}
void orc_vhm_add(void* m, const float* pts, const float* covs, const float* rgb, const float* inten, size_t n, const float* pose16) {
    static_cast<VoxelHashMap*>(m)->add_point_cloud(pts, covs, rgb, inten, n, pose16);
}
size_t orc_vhm_downsampling(void* m, const float* center3, float distance, float* pts_out, float* cov_out, float* rgb_out,
                            float* inten_out, uint64_t* keys_out) {
    return static_cast<VoxelHashMap*>(m)->downsampling(center3, distance, pts_out, cov_out, rgb_out, inten_out, keys_out);
}
float orc_vhm_overlap_ratio(void* m, const float* pts, size_t n, const float* pose16) {
    return static_cast<VoxelHashMap*>(m)->overlap_ratio(pts, n, pose16);
}
void orc_vhm_remove_old_data(void* m) { static_cast<VoxelHashMap*>(m)->remove_old_data(); }
void orc_log_spd3(const float* A9_colmajor, float* out9) {
    Mat3 A; std::memcpy(A.d, A9_colmajor, 36);
    const Mat3 r = log_spd_3x3(A); std::memcpy(out9, r.d, 36);
}
void orc_exp_spd3(const float* A9_colmajor, float* out9) {
    Mat3 A; std::memcpy(A.d, A9_colmajor, 36);
    const Mat3 r = exp_spd_3x3(A); std::memcpy(out9, r.d, 36);
}

}  // extern "C"
