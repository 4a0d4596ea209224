// deep_gemm_cpp -- the reference's pybind11 torch extension, rebuilt on the C ABI of libdga_hip.so.
// Same module name and the same three entry points as /root/reference/deep_gemm_ascend/framework/csrc/python_api.cpp:13-36
// (run_mmad_custom / run_mmad_rtc / run_mmad_bench over at::Tensor, void return, output written in place, current device
// stream, synchronous as gemm.hpp:110), plus the fp8 operators of BASELINE.json's north star.  torch types stay on this
// side of the boundary; everything below it is plain pointers and sizes (include/dga_hip.h).
// Built in-tree by deepgemm_ascend_amd/build_ext.py (hipcc, host code only -- there is no device code in this file).
#include <cstdlib>
#include <torch/extension.h>
#include <c10/hip/HIPStream.h>
#include <hip/hip_runtime_api.h>

#include <c10/core/DeviceGuard.h>

#include <initializer_list>
#include <string>
#include <tuple>

#include "dga_hip.h"

namespace {

void check(int rc, const char *what)   // DGA_HOST_ASSERT / DGAException (framework/csrc/utils/exception.hpp:9-33)
{
    TORCH_CHECK(rc == DGA_OK, what, ": ", dga_status_string(rc), rc == DGA_E_HIP ? " (hipError " : "",
                rc == DGA_E_HIP ? std::to_string(dga_last_hip_error()) + ")" : std::string());
}
void *cur_stream() { return c10::hip::getCurrentHIPStream().stream(); }   // gemm.hpp:72: the current device stream
void sync_stream() { TORCH_CHECK(hipStreamSynchronize(static_cast<hipStream_t>(cur_stream())) == hipSuccess, "stream sync"); }
void on_device(const at::Tensor &t, const char *name)
{
    TORCH_CHECK(t.is_cuda(), name, " must live on a HIP device (there is no CPU path)");
    TORCH_CHECK(t.is_contiguous(), name, " must be contiguous");
}
int dt16(const at::Tensor &t)
{
    TORCH_CHECK(t.scalar_type() == at::kBFloat16 || t.scalar_type() == at::kHalf, "expected bfloat16 / float16");
    return t.scalar_type() == at::kBFloat16 ? DGA_DT_BF16 : DGA_DT_FP16;
}
at::Tensor scratch(const at::Tensor &like, size_t bytes)   // caller-owned workspace: the binding plays the op runtime
{
    return at::empty({static_cast<int64_t>(bytes ? bytes : 1)}, like.options().dtype(at::kByte));
}
// the arithmetic policy of an fp8 call (api.py ARITHMETIC_POLICIES): strict = true or policy = "strict" -> dispatchPolicyTag 3;
// "bf16_exact" -> 7; "fast" -> whatever schedule the fast tiling names (-1); "fast_ue8m0" -> that schedule | 16 (-2);
// "bf16_exact_ue8m0" -> 7 | 16; "" -> the operator's default: $DGA_DEFAULT_POLICY, else bf16-exact (inside the 2-ULP contract)
int policy_tag(bool strict, const std::string &policy_in)
{
    std::string policy = policy_in;
    if (policy.empty() && !strict) {   // the library's default, parsed and validated once in the C library (dga_default_policy)
        char name[32] = {0};
        TORCH_CHECK(dga_default_policy(name, sizeof name) == DGA_OK, "$DGA_DEFAULT_POLICY names no arithmetic policy");
        policy = name;
        TORCH_CHECK(policy != "auto", "$DGA_DEFAULT_POLICY=auto is a policy of the Python API (deepgemm_ascend_amd.api); name one here");
    }
    TORCH_CHECK(policy.empty() || policy == "fast" || policy == "bf16_exact" || policy == "strict" || policy == "fast_ue8m0" ||
                    policy == "bf16_exact_ue8m0",
                "policy must be one of 'fast', 'bf16_exact', 'strict', 'fast_ue8m0', 'bf16_exact_ue8m0'");
    TORCH_CHECK(!(strict && !policy.empty() && policy != "strict"), "strict=True contradicts policy='", policy, "'");
    if (strict || policy == "strict") return DGA_POLICY_STRICT;
    if (policy == "bf16_exact") return DGA_POLICY_BF16_EXACT;
    if (policy == "bf16_exact_ue8m0") return DGA_POLICY_BF16_EXACT | DGA_POLICY_UE8M0_SCALES;
    if (policy == "fast_ue8m0") return -2;
    return -1;
}
dga_tiling_t tiling_for(int m, int n, int k, int groups, int expected_m, unsigned flags, int tag)
{
    dga_problem_t p{};
    p.m = m; p.n = n; p.k = k; p.groups = groups; p.expected_m = expected_m;
    p.layoutTagA = DGA_LAYOUT_ROW_MAJOR; p.layoutTagB = DGA_LAYOUT_COLUMN_MAJOR; p.layoutTagC = DGA_LAYOUT_ROW_MAJOR;
    p.dtype = DGA_DT_FP8_E4M3FN; p.flags = flags;
    dga_tiling_t t{};
    if (tag >= 0 && (tag & 7) == DGA_POLICY_BF16_EXACT) {   // that policy's own tile / split-K pick
        check(dga_tiling_bf16_exact(&p, &t), "tiling_bf16_exact");
        t.dispatchPolicyTag = static_cast<uint8_t>(tag);
        return t;
    }
    check(dga_tiling(&p, &t), "tiling");
    if (tag >= 0) t.dispatchPolicyTag = static_cast<uint8_t>(tag);
    if (tag == -2) t.dispatchPolicyTag |= DGA_POLICY_UE8M0_SCALES;
    return t;
}
// Operand checks of the fp8 operators -- the same the ctypes mirror makes (api.py _require): full shapes, dtypes, contiguity,
// one device.  Shapes and dtypes are checked BEFORE the device (so that a host-side test can see them), nothing is launched
// unless every check passed: an undersized out or scale tensor would otherwise be an out-of-bounds device access.
void want_shape(const at::Tensor &t, std::initializer_list<int64_t> shape, const char *name)
{
    TORCH_CHECK(t.dim() == static_cast<int64_t>(shape.size()), name, " must have rank ", shape.size(), ", got ", t.dim());
    int64_t i = 0;
    for (int64_t d : shape) {
        TORCH_CHECK(t.size(i) == d, name, " must be ", at::IntArrayRef(shape), ", got ", t.sizes());
        ++i;
    }
}
void want_fp8(const at::Tensor &t, const char *name)
{
    TORCH_CHECK(t.scalar_type() == at::kFloat8_e4m3fn || t.scalar_type() == at::kByte, name,
                " must be float8_e4m3fn or uint8 bytes, got ", t.scalar_type());
}
void want_dtype(const at::Tensor &t, at::ScalarType st, const char *name)
{
    TORCH_CHECK(t.scalar_type() == st, name, " must be ", st, ", got ", t.scalar_type());
}
void same_device(std::initializer_list<const at::Tensor *> ts)
{
    const at::Tensor *first = *ts.begin();
    for (const at::Tensor *t : ts) {
        on_device(*t, "operand");
        TORCH_CHECK(t->device() == first->device(), "all operands must live on one device");
    }
}

// ---- the reference's three entry points -------------------------------------------------------------------------
void run_mmad_rtc(const at::Tensor &x, const at::Tensor &y, at::Tensor &z)   // gemm.hpp:68-111
{
    on_device(x, "x"); on_device(y, "y"); on_device(z, "z");
    TORCH_CHECK(x.dim() == 3 && y.dim() == 3 && z.dim() == 3 && z.scalar_type() == at::kFloat, "x [B,M,K], y [B,K,N], z [B,M,N] f32");
    const int batch = x.size(0), m = x.size(1), k = y.size(1), n = y.size(2);
    TORCH_CHECK(x.size(2) == k && y.size(0) == batch && z.size(0) == batch && z.size(1) == m && z.size(2) == n, "shape mismatch");
    TORCH_CHECK(x.device() == y.device() && x.device() == z.device(), "all operands must live on one device");
    const c10::OptionalDeviceGuard guard(at::device_of(z));
    const size_t wsb = dga_mmad_workspace_bytes(batch, m, n, k, x.data_ptr());
    at::Tensor ws = scratch(x, wsb);
    check(dga_run_mmad_rtc_ws(x.data_ptr(), y.data_ptr(), z.data_ptr<float>(), batch, m, n, k, dt16(x),
                              wsb ? ws.data_ptr() : nullptr, wsb, cur_stream()), "run_mmad_rtc");
    sync_stream();                                                             // gemm.hpp:110
}

void run_mmad_bench(const at::Tensor &x, const at::Tensor &y, at::Tensor &z, at::Tensor &params)   // gemm_bench.hpp:49-113
{
    on_device(x, "x"); on_device(y, "y"); on_device(z, "z");
    TORCH_CHECK(x.dim() == 2 && y.dim() == 2 && z.dim() == 2 && z.scalar_type() == at::kFloat, "x [M,K], y [K,N], z [M,N] f32");
    TORCH_CHECK(params.scalar_type() == at::kInt && params.numel() == 28, "params must be int32[28]");
    const int m = x.size(0), k = y.size(0), n = y.size(1);
    TORCH_CHECK(x.size(1) == k && z.size(0) == m && z.size(1) == n, "shape mismatch");
    TORCH_CHECK(x.device() == y.device() && x.device() == z.device(), "all operands must live on one device");
    const c10::OptionalDeviceGuard guard(at::device_of(z));
    at::Tensor host = params.to(at::kCPU).contiguous();                        // the reference does 6 .item() syncs (:52-57)
    check(dga_bench_params_fill(m, n, k, host.data_ptr<int32_t>()), "bench_params_fill");
    params.copy_(host);                                                        // write-back of slots 6..27 (:68-81)
    const size_t wsb = dga_mmad_workspace_bytes(1, m, n, k, x.data_ptr());
    at::Tensor ws = scratch(x, wsb);
    check(dga_run_mmad_bench_ws(x.data_ptr(), y.data_ptr(), z.data_ptr<float>(), m, n, k, dt16(x),
                                host.data_ptr<int32_t>(), wsb ? ws.data_ptr() : nullptr, wsb, cur_stream()), "run_mmad_bench");
    sync_stream();
}

// ---- the fp8 block-scaled operators (names from upstream DeepGEMM) -----------------------------
void gemm_fp8_fp8_bf16_nt(const at::Tensor &a, const at::Tensor &sfa, const at::Tensor &b, const at::Tensor &sfb,
                          at::Tensor &out, bool strict, const std::string &policy)
{
    const int tag = policy_tag(strict, policy);
    TORCH_CHECK(a.dim() == 2 && b.dim() == 2, "a [M,K], b [N,K]");
    const int64_t m = a.size(0), n = b.size(0), k = a.size(1), kb = (k + 127) / 128, nb = (n + 127) / 128;
    want_fp8(a, "a"); want_fp8(b, "b");
    want_shape(b, {n, k}, "b");
    want_shape(out, {m, n}, "out"); want_dtype(out, at::kBFloat16, "out");
    want_shape(sfa, {m, kb}, "sfa"); want_dtype(sfa, at::kFloat, "sfa");
    want_shape(sfb, {nb, kb}, "sfb"); want_dtype(sfb, at::kFloat, "sfb");
    same_device({&a, &sfa, &b, &sfb, &out});
    const c10::OptionalDeviceGuard guard(at::device_of(out));   // stream, tiling (CU count) and scratch on the tensors' device
    const dga_tiling_t t = tiling_for(m, n, k, 1, 0, 0, tag);
    const size_t wsb = dga_workspace_bytes(&t);
    at::Tensor ws = scratch(out, wsb);
    check(dga_gemm_fp8_fp8_bf16_nt(a.data_ptr(), sfa.data_ptr<float>(), b.data_ptr(), sfb.data_ptr<float>(), out.data_ptr(),
                                   m, n, k, &t, wsb ? ws.data_ptr() : nullptr, wsb, cur_stream()), "gemm_fp8_fp8_bf16_nt");
}

void m_grouped_gemm_fp8_fp8_bf16_nt_masked(const at::Tensor &a, const at::Tensor &sfa, const at::Tensor &b,
                                           const at::Tensor &sfb, at::Tensor &out, const at::Tensor &masked_m,
                                           int64_t expected_m, bool strict, const std::string &policy)
{
    const int tag = policy_tag(strict, policy);
    TORCH_CHECK(a.dim() == 3 && b.dim() == 3, "a [G,Mmax,K], b [G,N,K]");
    const int64_t g = a.size(0), mmax = a.size(1), n = b.size(1), k = a.size(2), kb = (k + 127) / 128, nb = (n + 127) / 128;
    want_fp8(a, "a"); want_fp8(b, "b");
    want_shape(b, {g, n, k}, "b");
    want_shape(out, {g, mmax, n}, "out"); want_dtype(out, at::kBFloat16, "out");
    want_shape(sfa, {g, mmax, kb}, "sfa"); want_dtype(sfa, at::kFloat, "sfa");
    want_shape(sfb, {g, nb, kb}, "sfb"); want_dtype(sfb, at::kFloat, "sfb");
    want_shape(masked_m, {g}, "masked_m"); want_dtype(masked_m, at::kInt, "masked_m");
    same_device({&a, &sfa, &b, &sfb, &out, &masked_m});
    const c10::OptionalDeviceGuard guard(at::device_of(out));
    const dga_tiling_t t = tiling_for(mmax, n, k, g, static_cast<int>(expected_m), 0, tag);
    check(dga_m_grouped_gemm_fp8_fp8_bf16_nt_masked(a.data_ptr(), sfa.data_ptr<float>(), b.data_ptr(), sfb.data_ptr<float>(),
                                                    out.data_ptr(), masked_m.data_ptr<int32_t>(), g, mmax, n, k,
                                                    static_cast<int>(expected_m), &t, nullptr, 0, cur_stream()),
          "m_grouped_gemm_fp8_fp8_bf16_nt_masked");
}

void m_grouped_gemm_fp8_fp8_bf16_nt_contiguous(const at::Tensor &a, const at::Tensor &sfa, const at::Tensor &b,
                                               const at::Tensor &sfb, at::Tensor &out, const at::Tensor &m_indices, bool strict,
                                               const std::string &policy)
{
    const int tag = policy_tag(strict, policy);
    TORCH_CHECK(a.dim() == 2 && b.dim() == 3, "a [Msum,K], b [G,N,K]");
    const int64_t msum = a.size(0), g = b.size(0), n = b.size(1), k = a.size(1), kb = (k + 127) / 128, nb = (n + 127) / 128;
    want_fp8(a, "a"); want_fp8(b, "b");
    want_shape(b, {g, n, k}, "b");
    want_shape(out, {msum, n}, "out"); want_dtype(out, at::kBFloat16, "out");
    want_shape(sfa, {msum, kb}, "sfa"); want_dtype(sfa, at::kFloat, "sfa");
    want_shape(sfb, {g, nb, kb}, "sfb"); want_dtype(sfb, at::kFloat, "sfb");
    want_shape(m_indices, {msum}, "m_indices"); want_dtype(m_indices, at::kInt, "m_indices");
    same_device({&a, &sfa, &b, &sfb, &out, &m_indices});
    const c10::OptionalDeviceGuard guard(at::device_of(out));
    const dga_tiling_t t = tiling_for(msum, n, k, g, 0, DGA_PROBLEM_CONTIGUOUS_M, tag);
    const size_t wsb = dga_workspace_bytes(&t);
    at::Tensor ws = scratch(out, wsb);
    check(dga_m_grouped_gemm_fp8_fp8_bf16_nt_contiguous(a.data_ptr(), sfa.data_ptr<float>(), b.data_ptr(),
                                                        sfb.data_ptr<float>(), out.data_ptr(), m_indices.data_ptr<int32_t>(),
                                                        msum, g, n, k, &t, wsb ? ws.data_ptr() : nullptr, wsb, cur_stream()),
          "m_grouped_gemm_fp8_fp8_bf16_nt_contiguous");
}

std::tuple<at::Tensor, at::Tensor> cast_to_fp8(const at::Tensor &x, int block_rows)
{
    on_device(x, "x");
    TORCH_CHECK(x.dim() == 2, "x must be [rows, k]");
    const at::ScalarType st = x.scalar_type();
    TORCH_CHECK(st == at::kFloat || st == at::kBFloat16 || st == at::kHalf, "x must be float32 / bfloat16 / float16");
    const c10::OptionalDeviceGuard guard(at::device_of(x));
    at::Tensor q = at::empty(x.sizes(), x.options().dtype(at::kFloat8_e4m3fn));
    at::Tensor sf = at::empty({(x.size(0) + block_rows - 1) / block_rows, (x.size(1) + 127) / 128}, x.options().dtype(at::kFloat));
    const int dt = st == at::kFloat ? DGA_DT_FP32 : st == at::kBFloat16 ? DGA_DT_BF16 : DGA_DT_FP16;
    check((block_rows == 1 ? dga_cast_to_fp8_1x128 : dga_cast_to_fp8_128x128)(x.data_ptr(), dt, x.size(0), x.size(1), q.data_ptr(),
                                                                               sf.data_ptr<float>(), cur_stream()),
          "cast_to_fp8");
    return {q, sf};
}

}  // namespace

PYBIND11_MODULE(deep_gemm_cpp, m)   // the reference's module name (python_api.cpp:30)
{
    m.doc() = "MI355X drop-in for deep_gemm_ascend's deep_gemm_cpp (C ABI: include/dga_hip.h)";
    m.def("run_mmad_custom", [](const at::Tensor &, const at::Tensor &, at::Tensor &) {},
          "the reference's static kernel returns at once (include/impls/mmad.cpp:79): a no-op, kept for API parity");
    m.def("run_mmad_rtc", &run_mmad_rtc, "run_mmad_rtc");
    m.def("run_mmad_bench", &run_mmad_bench, "run_mmad_bench");
    m.def("gemm_fp8_fp8_bf16_nt", &gemm_fp8_fp8_bf16_nt, py::arg("a"), py::arg("sfa"), py::arg("b"), py::arg("sfb"),
          py::arg("out"), py::arg("strict") = false, py::arg("policy") = "");
    m.def("m_grouped_gemm_fp8_fp8_bf16_nt_masked", &m_grouped_gemm_fp8_fp8_bf16_nt_masked, py::arg("a"), py::arg("sfa"),
          py::arg("b"), py::arg("sfb"), py::arg("out"), py::arg("masked_m"), py::arg("expected_m"), py::arg("strict") = false,
          py::arg("policy") = "");
    m.def("m_grouped_gemm_fp8_fp8_bf16_nt_contiguous", &m_grouped_gemm_fp8_fp8_bf16_nt_contiguous, py::arg("a"),
          py::arg("sfa"), py::arg("b"), py::arg("sfb"), py::arg("out"), py::arg("m_indices"), py::arg("strict") = false,
          py::arg("policy") = "");
    m.def("get_m_alignment_for_contiguous_layout", [] { return DGA_CONTIGUOUS_M_ALIGNMENT; });
    m.def("per_token_cast_to_fp8", [](const at::Tensor &x) { return cast_to_fp8(x, 1); });
    m.def("per_block_cast_to_fp8", [](const at::Tensor &x) { return cast_to_fp8(x, 128); });
    m.def("abi_version", [] { return dga_abi_version(); });
}
