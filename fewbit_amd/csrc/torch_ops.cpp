// torch_ops.cpp -- the operator-registration boundary: TORCH_LIBRARY(fewbit) with the reference's
// schemas and AutogradCUDA implementations that call the gfx950 C-ABI (include/fewbit_hip.h) on
// torch's current HIP stream.  Built into fewbit_amd/libfewbit.so and loaded with
// torch.ops.load_library, exactly like the reference's fewbit/libfewbit.so (fewbit/__init__.py:17-23).
//
// What this file replaces in the reference (skolai/fewbit):
//   schemas ................ TORCH_LIBRARY(fewbit, m), fewbit/fewbit.cc:5-39 (same 24 names/signatures)
//   autograd Functions ..... fewbit/cuda/activation.cc:23-382 (8 hand-written + ContinousCudaFunction<T>)
//   impl registration ...... TORCH_LIBRARY_IMPL(fewbit, AutogradCUDA, m), fewbit/cuda/activation.cc:445-470
//   quantize(_backward) .... fewbit/cpu/gelu.cc:7-45 (there CPU-only; here the same two raw ops on the GPU)
// Differences, all deliberate: bit width is ceil(log2(#levels)) (defect 1 not reproduced),
// kernels run on the current stream and launch errors surface as exceptions (defect 9), fp16/bf16 are
// accepted besides fp32, inputs are checked (contiguity, device, dtype) instead of silently mis-indexed.
//
// Dispatch keys:
//   AutogradCUDA  all activations: autograd Function around the gfx950 kernels (as the reference)
//   CUDA          the same launches without an autograd node (torch.inference_mode(), where the Autograd keys are
//                 excluded); quantize / quantize_backward
//   AutogradCPU   host tensors: this file's own ATen-level implementation with the same PACKED state (section "host
//                 tensors" below), layout-agnostic like the reference's host operator.  The reference registers `gelu`
//                 only, as an autograd Function under the CPU key (fewbit/cpu/gelu.cc:47-76), plus quantize /
//                 quantize_backward (fewbit/fewbit.cc:6-7); here every operator has one.  In-place rule on the host:
//                 `gelu` returns a FRESH tensor and leaves its input intact -- that is what the reference's host operator
//                 does whatever its schema says (fewbit/cpu/gelu.cc:7-31), pinned by tests/test_host_ops.py; every OTHER
//                 operator (no reference host implementation to match) honours its `Tensor(a!)` schema: the result is
//                 written into `self`, which is returned, exactly as on the GPU.  Product code; it never touches oracle/.
//   CPU           the same without an autograd node (plain activation), quantize / quantize_backward
//
// Autograd plumbing, two routes with identical results (tests: both give the same bytes and gradients):
//   direct    one hand-written torch::autograd::Node per call (PackedBackwardNode below) wired the way ATen's generated
//             code wires its own (check_inplace / collect_next_edges / rebase_history / set_history): ~3 us less host
//             time per forward+backward than a torch::autograd::Function.  These helpers -- and the view meta data that
//             `whole_view_base` reads -- are internal headers of libtorch, so the route is compiled only for the torch
//             release it was written and tested against (FEWBIT_AUTOGRAD_INTERNALS, see below).
//   portable  torch::autograd::Function (public API only): every other torch release, -DFEWBIT_AUTOGRAD_INTERNALS=0, or at
//             run time FEWBIT_NO_DIRECT_NODE=1 / FEWBIT_NO_BASE_DIRTY=1 (or fewbit_torch_route(), bottom of this file).
#include <ATen/OpMathType.h>
#include <ATen/Parallel.h>
#include <torch/csrc/autograd/variable.h>
#include <torch/library.h>
#include <torch/torch.h>

#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>

// The internal-API routes are tied to the release they were verified against: another torch falls back to the public
// API at compile time instead of building against headers that may have changed meaning.
#ifndef FEWBIT_AUTOGRAD_INTERNALS
#if defined(TORCH_VERSION_MAJOR) && TORCH_VERSION_MAJOR == 2 && TORCH_VERSION_MINOR == 10
#define FEWBIT_AUTOGRAD_INTERNALS 1
#else
#define FEWBIT_AUTOGRAD_INTERNALS 0
#endif
#endif
#if FEWBIT_AUTOGRAD_INTERNALS
#include <torch/csrc/autograd/VariableTypeUtils.h>     // check_inplace, rebase_history, increment_version
#include <torch/csrc/autograd/functions/utils.h>       // set_history, collect_next_edges
#endif

// ROCm builds of PyTorch expose HIP devices as device type `cuda`; these are the matching guard / stream types
#include <ATen/hip/EmptyTensor.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>

#include "fewbit_hip.h"

namespace fewbit_amd {

using torch::Tensor;
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

namespace {

int dtype_code(const Tensor &t) {
    switch (t.scalar_type()) {
    case torch::kFloat32: return FEWBIT_F32;
    case torch::kFloat16: return FEWBIT_F16;
    case torch::kBFloat16: return FEWBIT_BF16;
    default: TORCH_CHECK(false, "fewbit: unsupported dtype ", t.scalar_type(), " (expected float32, float16 or bfloat16)");
    }
}

void *current_stream(const Tensor &t) { return c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.get_device()).stream(); }

void check_status(int rc, const char *what) {
    TORCH_CHECK(rc == FEWBIT_OK, "fewbit: ", what, " failed (", rc, "): ", fewbit_hip_last_error());
}

void check_input(const Tensor &t, const char *name) {
    TORCH_CHECK(t.is_cuda(), "fewbit: `", name, "` must be a GPU tensor, got ", t.device());
    TORCH_CHECK(t.is_contiguous(), "fewbit: `", name, "` must be contiguous");
}

void check_table(const Tensor &self, const Tensor &table, const char *name) {
    TORCH_CHECK(table.dim() == 1, "fewbit: `", name, "` must be one-dimensional");
    TORCH_CHECK(table.device() == self.device(), "fewbit: `", name, "` lives on ", table.device(), ", input on ", self.device());
    TORCH_CHECK(table.scalar_type() == self.scalar_type(), "fewbit: `", name, "` has dtype ", table.scalar_type(),
                ", input ", self.scalar_type());
}

// the packed state: straight from the caching allocator (what at::empty ends in for a GPU tensor, without its two
// dispatcher hops -- this allocation is on the path of every forward)
Tensor new_state(const Tensor &like, int64_t numel, int nbits) {
    const auto nbytes = static_cast<int64_t>(fewbit_hip_state_nbytes(static_cast<size_t>(numel), nbits));
    return Tensor(at::detail::empty_cuda({nbytes}, torch::kUInt8, like.device(), std::nullopt));
}

// ---- raw launches (no autograd) ------------------------------------------------------------------

// y = fn(self) written to `out` (which may be `self` itself: in place); returns the packed state
Tensor launch_quantize(int fn, const Tensor &self, Tensor &out, const Tensor &bounds, double p0, double p1) {
    check_input(self, "self");
    check_table(self, bounds, "bounds");
    const Tensor b = bounds.contiguous();  // e.g. borders[1:-1] is already contiguous; strided views are not
    TORCH_CHECK(b.numel() >= 1 && b.numel() <= 255, "fewbit: number of borders must be in [1, 255], got ", b.numel());
    const int nbits = fewbit_hip_bitwidth(static_cast<int>(b.numel()) + 1);
    Tensor state = new_state(self, self.numel(), nbits);
    c10::hip::HIPGuardMasqueradingAsCUDA guard(self.device());
    check_status(fewbit_hip_quantize_forward(fn, dtype_code(self), self.data_ptr(), out.data_ptr(),
                                             state.data_ptr<uint8_t>(), static_cast<size_t>(self.numel()),
                                             b.data_ptr(), static_cast<int>(b.numel()), p0, p1, current_stream(self)),
                 "quantize_forward");
    return state;
}

Tensor launch_dequantize(const Tensor &grad, const Tensor &state, const Tensor &levels) {
    Tensor gy = grad.contiguous();
    check_input(gy, "grad_output");
    check_table(gy, levels, "levels");
    const Tensor lv = levels.contiguous();
    TORCH_CHECK(lv.numel() >= 2 && lv.numel() <= 256, "fewbit: number of levels must be in [2, 256], got ", lv.numel());
    const int nbits = fewbit_hip_bitwidth(static_cast<int>(lv.numel()));
    TORCH_CHECK(state.is_cuda() && state.scalar_type() == torch::kUInt8 && state.is_contiguous(),
                "fewbit: state must be a contiguous uint8 GPU tensor");
    TORCH_CHECK(state.numel() >= static_cast<int64_t>(fewbit_hip_state_nbytes(static_cast<size_t>(gy.numel()), nbits)),
                "fewbit: state buffer too small for ", gy.numel(), " elements at ", nbits, " bits");
    Tensor gx = torch::empty_like(gy);
    c10::hip::HIPGuardMasqueradingAsCUDA guard(gy.device());
    check_status(fewbit_hip_quantize_backward(dtype_code(gy), gy.data_ptr(), state.data_ptr<uint8_t>(), gx.data_ptr(),
                                              static_cast<size_t>(gy.numel()), lv.data_ptr(),
                                              static_cast<int>(lv.numel()), current_stream(gy)),
                 "quantize_backward");
    return gx;
}

// 1-bit family: y = fn(self) into `out` (may be `self`); returns the packed bits
Tensor launch_step1(int fn, const Tensor &self, Tensor &out, double p0, double p1) {
    check_input(self, "self");
    Tensor state = new_state(self, self.numel(), 1);
    c10::hip::HIPGuardMasqueradingAsCUDA guard(self.device());
    check_status(fewbit_hip_stepwise1_forward(fn, dtype_code(self), self.data_ptr(), out.data_ptr(), state.data_ptr<uint8_t>(),
                                              static_cast<size_t>(self.numel()), p0, p1, current_stream(self)),
                 "stepwise1_forward");
    return state;
}

Tensor launch_step1_backward(int fn, const Tensor &grad, const Tensor &state, double p0) {
    Tensor gy = grad.contiguous();
    check_input(gy, "grad_output");
    Tensor gx = torch::empty_like(gy);
    c10::hip::HIPGuardMasqueradingAsCUDA guard(gy.device());
    check_status(fewbit_hip_stepwise1_backward(fn, dtype_code(gy), gy.data_ptr(), state.data_ptr<uint8_t>(), gx.data_ptr(),
                                               static_cast<size_t>(gy.numel()), p0, current_stream(gy)),
                 "stepwise1_backward");
    return gx;
}

// ---- run-time route switches (see the header comment) ---------------------------------------------------------------
namespace route {
enum Which { DirectNode, BaseDirty, FreshView, Count };
constexpr const char *kNames[Count] = {"direct_node", "base_dirty", "fresh_view"};
constexpr const char *kOffEnv[Count] = {"FEWBIT_NO_DIRECT_NODE", "FEWBIT_NO_BASE_DIRTY", "FEWBIT_NO_FRESH_VIEW"};
constexpr bool kNeedsInternals[Count] = {true, true, false};
std::atomic<int> g_state[Count] = {{-1}, {-1}, {-1}};      // -1: not read yet (environment decides on first use)

bool on(Which w) {
    if (kNeedsInternals[w] && !FEWBIT_AUTOGRAD_INTERNALS) return false;
    int v = g_state[w].load(std::memory_order_relaxed);
    if (v < 0) {
        const char *e = std::getenv(kOffEnv[w]);
        v = (e && *e && std::strcmp(e, "0") != 0) ? 0 : 1;
        g_state[w].store(v, std::memory_order_relaxed);
    }
    return v != 0;
}
}  // namespace route

// ---- autograd, portable route: torch::autograd::Function (public API only) ---------------------------------------------

// all 13 continuous activations (+ custom `stepwise` tables): state and levels are what is saved
struct ContinuousFunction : public torch::autograd::Function<ContinuousFunction> {
    static Tensor forward(AutogradContext *ctx, Tensor self, const Tensor &bounds, const Tensor &levels, int64_t fn,
                          double p0, double p1, bool inplace) {
        Tensor out = inplace ? self : torch::empty_like(self);
        Tensor state = launch_quantize(static_cast<int>(fn), self, out, bounds, p0, p1);
        if (inplace) ctx->mark_dirty({self});
        ctx->save_for_backward({state, levels});
        return out;
    }

    static variable_list backward(AutogradContext *ctx, variable_list grad_output) {
        const auto saved = ctx->get_saved_variables();
        return {launch_dequantize(grad_output[0], saved[0], saved[1]), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(),
                Tensor()};
    }
};

// the eight piecewise-linear activations with an exact 1-bit state
struct Stepwise1Function : public torch::autograd::Function<Stepwise1Function> {
    static Tensor forward(AutogradContext *ctx, Tensor self, int64_t fn, double p0, double p1, bool inplace) {
        check_input(self, "self");
        Tensor out = inplace ? self : torch::empty_like(self);
        Tensor state = launch_step1(static_cast<int>(fn), self, out, p0, p1);
        if (inplace) ctx->mark_dirty({self});
        ctx->save_for_backward({state});
        ctx->saved_data["fn"] = fn;
        ctx->saved_data["p0"] = p0;
        return out;
    }

    static variable_list backward(AutogradContext *ctx, variable_list grad_output) {
        const auto saved = ctx->get_saved_variables();
        const auto fn = ctx->saved_data["fn"].toInt();
        const auto p0 = ctx->saved_data["p0"].toDouble();
        return {launch_step1_backward(static_cast<int>(fn), grad_output[0], saved[0], p0), Tensor(), Tensor(), Tensor(), Tensor()};
    }
};

// ---- host tensors (CPU dispatch key) ---------------------------------------------------------------
// Own ATen-level implementation of the same path for host tensors, with the same PACKED state as the kernels (k
// bytes per group of 8 elements, k*ceil(n/8) bytes, padding codes zero) -- so a state produced on the host can be
// moved to the GPU and consumed there, and the reverse.  Structure after fewbit/cpu/gelu.cc:7-65 (activation by
// ATen, codes by torch::searchsorted -- NaN -> last code --, pack, {state, levels} saved, unpack + gather + multiply in
// backward); the pack/unpack loops are this file's own (per group, parallel over groups) instead of the reference's
// serial bit-stream walk (fewbit/cpu/codec.h:33-83), with identical bytes.

int bitwidth_of(int64_t nlevels) { return fewbit_hip_bitwidth(static_cast<int>(nlevels)); }

// int32 codes (one per element) -> packed state
Tensor host_pack(const Tensor &codes, int nbits) {
    const int64_t n = codes.numel(), groups = (n + 7) / 8;
    Tensor state = torch::empty({static_cast<int64_t>(nbits) * groups}, torch::TensorOptions().dtype(torch::kUInt8));
    const int32_t *c = codes.data_ptr<int32_t>();
    uint8_t *out = state.data_ptr<uint8_t>();
    const uint64_t mask = (1ull << nbits) - 1ull;
    at::parallel_for(0, groups, 2048, [&](int64_t g0, int64_t g1) {
        for (int64_t g = g0; g < g1; ++g) {
            const int64_t e0 = g * 8, m = std::min<int64_t>(8, n - e0);
            uint64_t w = 0;
            for (int64_t i = 0; i < m; ++i) w |= (static_cast<uint64_t>(static_cast<uint32_t>(c[e0 + i])) & mask) << (nbits * i);
            for (int j = 0; j < nbits; ++j) out[nbits * g + j] = static_cast<uint8_t>(w >> (8 * j));
        }
    });
    return state;
}

// gx[i] = table[code_i] * gy[i]: product in the op-math type (fp32 for 16-bit tensors), rounded once to nearest even
// (`pair`: the two fp32 multipliers of the 1-bit family instead of a level table in the tensor dtype)
Tensor host_unpack_mul(const Tensor &grad, const Tensor &state, const Tensor &table, int nbits, const float *pair = nullptr) {
    const Tensor gy = grad.contiguous();
    TORCH_CHECK(gy.is_floating_point(), "fewbit: unsupported gradient dtype ", gy.scalar_type());
    if (!pair) {
        TORCH_CHECK(table.defined() && table.device().is_cpu() && table.is_contiguous() && table.scalar_type() == gy.scalar_type(),
                    "fewbit: `levels` must be a contiguous host tensor of the gradient's dtype (", gy.scalar_type(), ")");
    }
    TORCH_CHECK(state.device().is_cpu() && state.scalar_type() == torch::kUInt8 && state.is_contiguous(),
                "fewbit: state must be a contiguous uint8 host tensor");
    const int64_t n = gy.numel(), groups = (n + 7) / 8;
    TORCH_CHECK(state.numel() >= (static_cast<int64_t>(nbits) * n + 7) / 8, "fewbit: state buffer too small for ", n,
                " elements at ", nbits, " bits");
    const int64_t nstate = state.numel(), nlev = pair ? 2 : table.numel();
    Tensor gx = torch::empty_like(gy);
    const uint8_t *in = state.data_ptr<uint8_t>();
    const uint64_t mask = (1ull << nbits) - 1ull;
    AT_DISPATCH_FLOATING_TYPES_AND2(torch::kHalf, torch::kBFloat16, gy.scalar_type(), "fewbit_host_backward", [&] {
        using acc_t = at::opmath_type<scalar_t>;
        const scalar_t *g = gy.data_ptr<scalar_t>(), *lv = pair ? nullptr : table.data_ptr<scalar_t>();
        scalar_t *o = gx.data_ptr<scalar_t>();
        at::parallel_for(0, groups, 2048, [&](int64_t g0, int64_t g1) {
            for (int64_t gr = g0; gr < g1; ++gr) {
                uint64_t w = 0;
                for (int j = 0; j < nbits; ++j)     // (a reference-sized state, ceil(k*n/8) bytes, ends inside the last group)
                    if (nbits * gr + j < nstate) w |= static_cast<uint64_t>(in[nbits * gr + j]) << (8 * j);
                const int64_t e0 = gr * 8, m = std::min<int64_t>(8, n - e0);
                for (int64_t i = 0; i < m; ++i) {
                    const auto code = (w >> (nbits * i)) & mask;
                    // (a code beyond the table -- only a foreign or corrupted state has one -- reads as level 0, as in the kernels)
                    const acc_t level = pair ? static_cast<acc_t>(pair[code])
                                             : (static_cast<int64_t>(code) < nlev ? static_cast<acc_t>(lv[code]) : acc_t(0));
                    o[e0 + i] = static_cast<scalar_t>(level * static_cast<acc_t>(g[e0 + i]));
                }
            }
        });
    });
    return gx;
}

// the plain ATen activation behind each continuous id (what the reference's fallback meant to call,
// fewbit/functional/activations.py:107,253-261; for gelu what its native CPU op calls, fewbit/cpu/gelu.cc:12-16)
Tensor host_activation(int fn, const Tensor &x, double p0, double p1) {
    switch (fn) {
    case FEWBIT_CELU: return torch::celu(x, p0);
    case FEWBIT_ELU: return torch::elu(x, p0);
    case FEWBIT_GELU: return torch::gelu(x);
    case FEWBIT_HARDSWISH: return torch::hardswish(x);
    case FEWBIT_LOGSIGMOID: return torch::log_sigmoid(x);
    case FEWBIT_MISH: return torch::mish(x);
    case FEWBIT_SELU: return torch::selu(x);
    case FEWBIT_SIGMOID: return torch::sigmoid(x);
    case FEWBIT_SILU: return torch::silu(x);
    case FEWBIT_SOFTPLUS: return torch::softplus(x, p0, p1);
    case FEWBIT_SOFTSIGN: return x / (x.abs() + 1);
    case FEWBIT_TANH: return torch::tanh(x);
    case FEWBIT_TANHSHRINK: return x - torch::tanh(x);
    default: return x.clone();    // FEWBIT_IDENTITY, FEWBIT_IDENTITY_FOLD
    }
}

void check_host_table(const Tensor &self, const Tensor &table, const char *name) {
    TORCH_CHECK(table.dim() == 1, "fewbit: `", name, "` must be one-dimensional");
    TORCH_CHECK(table.device().is_cpu(), "fewbit: `", name, "` lives on ", table.device(), ", input on ", self.device());
    TORCH_CHECK(table.scalar_type() == self.scalar_type(), "fewbit: `", name, "` has dtype ", table.scalar_type(),
                ", input ", self.scalar_type());
}

// y = fn(self), written into `out` when `out` is defined (in place: `out` is `self`) and as a fresh tensor otherwise;
// any layout is accepted: codes and state follow the logical row-major order (hard part 7: what the reference's
// CPU path does for a strided input, fewbit/cpu/gelu.cc:7-31).  Returns the packed state.
Tensor host_quantize(int fn, const Tensor &self, Tensor &out, const Tensor &bounds, double p0, double p1) {
    TORCH_CHECK(self.is_floating_point(), "fewbit: unsupported dtype ", self.scalar_type());
    check_host_table(self, bounds, "bounds");
    const Tensor b = bounds.contiguous();
    TORCH_CHECK(b.numel() >= 1 && b.numel() <= 255, "fewbit: number of borders must be in [1, 255], got ", b.numel());
    const int nbits = bitwidth_of(b.numel() + 1);
    const Tensor x = self.contiguous();
    const Tensor flat = x.reshape({-1});
    // the even-parity fold of a custom table searches the fp32 distance |x - shift_x| (see stepwise_folded)
    const Tensor codes = fn == FEWBIT_IDENTITY_FOLD
                             ? torch::searchsorted(b.to(torch::kFloat), flat.to(torch::kFloat).sub(p0).abs(), /*out_int32=*/true)
                             : torch::searchsorted(b, flat, /*out_int32=*/true);
    Tensor state = host_pack(codes, nbits);
    Tensor y = host_activation(fn, x, p0, p1);
    if (out.defined())
        out.copy_(y);
    else
        out = y;
    return state;
}

Tensor host_dequantize(const Tensor &grad, const Tensor &state, const Tensor &levels) {
    const Tensor lv = levels.contiguous();
    return host_unpack_mul(grad, state, lv, bitwidth_of(lv.numel()));
}

// 1-bit family on the host: values by ATen, the derivative-branch bit by the same rules as the kernels
// (Step1<> in fewbit_device.h; fewbit/cuda/codec.cu:298-487)
Tensor host_step1_activation(int fn, const Tensor &x, double p0, double p1) {
    switch (fn) {
    case FEWBIT_HARDSHRINK: return torch::hardshrink(x, p0);
    case FEWBIT_HARDSIGMOID: return torch::hardsigmoid(x);
    case FEWBIT_HARDTANH: return torch::hardtanh(x, p0, p1);
    case FEWBIT_LEAKY_RELU: return torch::leaky_relu(x, p0);
    case FEWBIT_RELU: return torch::relu(x);
    case FEWBIT_RELU6: return torch::relu6(x);
    case FEWBIT_SOFTSHRINK: return torch::softshrink(x, p0);
    default: return torch::threshold(x, p0, p1);
    }
}

Tensor host_step1_bits(int fn, const Tensor &x, double p0, double p1) {
    const Tensor v = x.reshape({-1}).to(torch::kFloat);
    const float a = static_cast<float>(p0), b = static_cast<float>(p1);
    Tensor bit;
    switch (fn) {
    case FEWBIT_HARDSHRINK: bit = (v < -a).logical_or(v > a); break;
    case FEWBIT_HARDSIGMOID: bit = (v <= -3.0f).logical_or(v >= 3.0f).logical_not(); break;
    case FEWBIT_HARDTANH: bit = (v <= a).logical_or(v >= b).logical_not(); break;
    case FEWBIT_LEAKY_RELU: bit = (v >= 0.0f).logical_not(); break;
    case FEWBIT_RELU: bit = (v <= 0.0f).logical_not(); break;
    case FEWBIT_RELU6: bit = (v <= 0.0f).logical_or(v >= 6.0f).logical_not(); break;
    case FEWBIT_SOFTSHRINK: bit = (v < -a).logical_or(v > a); break;
    default: bit = (v <= a).logical_not(); break;
    }
    return bit.to(torch::kInt32);
}

// as host_quantize: `out` defined = write the values there (in place), else a fresh tensor
Tensor host_step1(int fn, const Tensor &self, Tensor &out, double p0, double p1) {
    TORCH_CHECK(self.is_floating_point(), "fewbit: unsupported dtype ", self.scalar_type());
    const Tensor x = self.contiguous();
    Tensor state = host_pack(host_step1_bits(fn, x, p0, p1), 1);
    Tensor y = host_step1_activation(fn, x, p0, p1);
    if (out.defined())
        out.copy_(y);
    else
        out = y;
    return state;
}

Tensor host_step1_backward(int fn, const Tensor &grad, const Tensor &state, double p0) {
    float pair[2] = {0.0f, 1.0f};       // as fewbit_hip_stepwise1_backward: (m0, m1) in fp32
    if (fn == FEWBIT_HARDSIGMOID) pair[1] = 1.0f / 6.0f;
    if (fn == FEWBIT_LEAKY_RELU) { pair[0] = 1.0f; pair[1] = static_cast<float>(p0); }
    return host_unpack_mul(grad, state, Tensor(), 1, pair);
}

struct HostContinuousFunction : public torch::autograd::Function<HostContinuousFunction> {
    static Tensor forward(AutogradContext *ctx, Tensor self, const Tensor &bounds, const Tensor &levels, int64_t fn,
                          double p0, double p1, bool inplace) {
        Tensor out = inplace ? self : Tensor();
        Tensor state = host_quantize(static_cast<int>(fn), self, out, bounds, p0, p1);
        if (inplace) ctx->mark_dirty({self});
        ctx->save_for_backward({state, levels});
        return out;
    }

    static variable_list backward(AutogradContext *ctx, variable_list grad_output) {
        const auto saved = ctx->get_saved_variables();
        return {host_dequantize(grad_output[0], saved[0], saved[1]), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
    }
};

struct HostStepwise1Function : public torch::autograd::Function<HostStepwise1Function> {
    static Tensor forward(AutogradContext *ctx, Tensor self, int64_t fn, double p0, double p1, bool inplace) {
        Tensor out = inplace ? self : Tensor();
        Tensor state = host_step1(static_cast<int>(fn), self, out, p0, p1);
        if (inplace) ctx->mark_dirty({self});
        ctx->save_for_backward({state});
        ctx->saved_data["fn"] = fn;
        ctx->saved_data["p0"] = p0;
        return out;
    }

    static variable_list backward(AutogradContext *ctx, variable_list grad_output) {
        const auto saved = ctx->get_saved_variables();
        const auto fn = ctx->saved_data["fn"].toInt();
        const auto p0 = ctx->saved_data["p0"].toDouble();
        return {host_step1_backward(static_cast<int>(fn), grad_output[0], saved[0], p0), Tensor(), Tensor(), Tensor(), Tensor()};
    }
};

// does this call need an autograd node at all?
bool needs_node(const Tensor &self) { return torch::GradMode::is_enabled() && self.requires_grad(); }

// an in-place write outside autograd still has to bump the tensor's version counter, so that a graph which saved the
// old value for another operand's gradient fails loudly in backward instead of using the overwritten data
void note_inplace_write(const Tensor &self) {
    if (!self.is_inference()) self.unsafeGetTensorImpl()->bump_version();
}

void check_table_sizes(const Tensor &bounds, const Tensor &levels) {
    TORCH_CHECK(bounds.numel() + 1 == levels.numel(),
                "fewbit: size of `bounds` should be lesser than size of `levels` by one, got ", bounds.numel(), " and ",
                levels.numel());
}

#if FEWBIT_AUTOGRAD_INTERNALS
// ---- autograd, direct route: one backward node, wired like ATen's generated code wires its own ---------------------
// (torch/csrc/autograd/generated/VariableType_*.cpp pattern: check_inplace -> grad_fn->set_next_edges(collect_next_edges)
// -> kernel below autograd -> increment_version + rebase_history (in place) / set_history (fresh result) -> SavedVariable.)
struct PackedBackwardNode : public torch::autograd::Node {
    torch::autograd::SavedVariable state, levels;     // `levels` stays empty for the 1-bit family
    int step_fn = -1;                                  // >= 0: a 1-bit operator
    double p0 = 0.0;
    bool host = false;

    variable_list apply(variable_list &&grads) override {
        std::lock_guard<std::mutex> lock(mutex_);      // (as ATen's generated nodes: apply and release_variables exclude each other)
        variable_list out(1);
        const Tensor st = state.unpack();              // raises autograd's own "backward through the graph a second time"
        const Tensor &g = grads[0];
        if (!g.defined() || !task_should_compute_output(0)) return out;
        if (step_fn >= 0) {
            out[0] = host ? host_step1_backward(step_fn, g, st, p0) : launch_step1_backward(step_fn, g, st, p0);
        } else {
            const Tensor lv = levels.unpack();
            out[0] = host ? host_dequantize(g, st, lv) : launch_dequantize(g, st, lv);
        }
        return out;
    }

    void release_variables() override {
        std::lock_guard<std::mutex> lock(mutex_);
        state.reset_data();
        levels.reset_data();
    }

    std::string name() const override { return "FewbitPackedBackward"; }
};

// `launch(out)` runs below autograd: it writes the values into `out` (defined = in place, `out` is `self`) or makes `out`,
// and returns the packed state
template <typename Launch>
Tensor forward_direct(const Tensor &self, bool inplace, bool host, int step_fn, double p0, const Tensor &levels, Launch &&launch) {
    if (inplace) torch::autograd::check_inplace(self, /*requires_grad=*/true);
    // forward-mode AD: the packed state holds what BACKWARD needs, there is no jvp -- say so (the torch::autograd::Function route
    // raises for a tangent too) instead of dropping the tangent silently
    TORCH_CHECK(!self._fw_grad(/*level=*/0).defined(), "fewbit: forward-mode AD (a tangent on the input) is not implemented for the packed-state operators");
    std::shared_ptr<PackedBackwardNode> node(new PackedBackwardNode(), torch::autograd::deleteNode);
    node->host = host;
    node->step_fn = step_fn;
    node->p0 = p0;
    node->set_next_edges(torch::autograd::collect_next_edges(self));
    Tensor out = inplace ? self : Tensor();
    Tensor state;
    {
        at::AutoDispatchBelowADInplaceOrView below;
        state = launch(out);
    }
    if (inplace) {
        torch::autograd::increment_version(self);
        torch::autograd::rebase_history(self, node);
    } else {
        torch::autograd::set_history(out, node);
    }
    node->state = torch::autograd::SavedVariable(state, /*is_output=*/false);
    if (levels.defined()) node->levels = torch::autograd::SavedVariable(levels, /*is_output=*/false);
    return out;
}
#endif

// In place on a VIEW that covers its whole base -- what the reference's callers do: the 3-D output of nn.Linear is a view
// of its 2-D addmm result, and benchmark/bench-roberta.py:138-147 hands it straight to torch.ops.fewbit.gelu.  Modifying
// the view itself makes autograd rebase it (CopySlices: a zero-fill plus three full-size copies around our backward).
// Modifying the BASE instead is the same write to the same memory: nothing is saved but {state, levels}, and
//   * the RETURNED tensor is a fresh `base.view(sizes)` (route `fresh_view`): its backward is a reshape, no copy at all;
//   * the tensor that was passed in stays correct for later users -- autograd re-derives its grad_fn from the base's new
//     one by itself (the standard "base modified after the view was taken" path: an AsStridedBackward0 node, whose
//     backward costs one zero-fill and one copy of the base -- only paid by graphs that keep using the OLD python object).
// Returns the undefined tensor when `self` is not such a view, or when the route is off / not compiled (then autograd's
// general in-place-on-view machinery runs: correct, slower).
Tensor whole_view_base(const Tensor &self) {
#if FEWBIT_AUTOGRAD_INTERNALS
    if (!self.is_view() || !route::on(route::BaseDirty)) return Tensor();
    // only ordinary views: for the kinds autograd refuses to modify in place (outputs of multi-output view ops, views made
    // under no_grad or inside a custom Function) the general route keeps raising autograd's own error
    const auto *meta = torch::autograd::impl::get_view_autograd_meta(self);
    if (!meta || meta->get_creation_meta() != torch::autograd::CreationMeta::DEFAULT) return Tensor();
    const Tensor base(self._base());      // (TensorBase::_base returns a const TensorBase &)
    if (!base.defined() || !base.requires_grad() || base.is_leaf() || !base.is_contiguous() || !self.is_contiguous() ||
        base.numel() != self.numel() || base.storage_offset() != self.storage_offset() || base.scalar_type() != self.scalar_type() ||
        base.device() != self.device())
        return Tensor();
    return base;
#else
    (void)self;
    return Tensor();
#endif
}

// what the caller gets back after the base was modified in its place
Tensor after_base_write(const Tensor &self, const Tensor &base) {
    return route::on(route::FreshView) ? base.view(self.sizes()) : self;
}

// ---- the four flavours every operator is registered in ---------------------------------------------
enum class Where { AutogradGpu, RawGpu, AutogradHost, RawHost };

// with an autograd node, device or host tensors, either route
Tensor continuous_with_node(bool host, int fn, const Tensor &self, const Tensor &bounds, const Tensor &levels, double p0,
                            double p1, bool inplace) {
#if FEWBIT_AUTOGRAD_INTERNALS
    if (route::on(route::DirectNode)) {
        return forward_direct(self, inplace, host, -1, 0.0, levels, [&](Tensor &out) {
            if (host) return host_quantize(fn, self, out, bounds, p0, p1);
            if (!out.defined()) out = torch::empty_like(self);
            return launch_quantize(fn, self, out, bounds, p0, p1);
        });
    }
#endif
    return host ? HostContinuousFunction::apply(self, bounds, levels, static_cast<int64_t>(fn), p0, p1, inplace)
                : ContinuousFunction::apply(self, bounds, levels, static_cast<int64_t>(fn), p0, p1, inplace);
}

Tensor stepwise1_with_node(bool host, int fn, const Tensor &self, double p0, double p1, bool inplace) {
#if FEWBIT_AUTOGRAD_INTERNALS
    if (route::on(route::DirectNode)) {
        return forward_direct(self, inplace, host, fn, p0, Tensor(), [&](Tensor &out) {
            if (host) return host_step1(fn, self, out, p0, p1);
            check_input(self, "self");
            if (!out.defined()) out = torch::empty_like(self);
            return launch_step1(fn, self, out, p0, p1);
        });
    }
#endif
    return host ? HostStepwise1Function::apply(self, static_cast<int64_t>(fn), p0, p1, inplace)
                : Stepwise1Function::apply(self, static_cast<int64_t>(fn), p0, p1, inplace);
}

template <Where W>
Tensor continuous(int fn, const Tensor &self, const Tensor &bounds, const Tensor &levels, double p0 = 0.0, double p1 = 0.0,
                  bool inplace = true) {
    check_table_sizes(bounds, levels);
    if constexpr (W == Where::AutogradGpu) {
        // nothing will ever ask for this call's gradient: skip the autograd node (and its host time)
        if (!needs_node(self)) return continuous<Where::RawGpu>(fn, self, bounds, levels, p0, p1, inplace);
        check_table(self, levels, "levels");
        if (inplace) {
            if (const Tensor base = whole_view_base(self); base.defined()) {
                continuous_with_node(false, fn, base, bounds, levels, p0, p1, true);
                return after_base_write(self, base);
            }
        }
        return continuous_with_node(false, fn, self, bounds, levels, p0, p1, inplace);
    } else if constexpr (W == Where::AutogradHost) {
        if (!needs_node(self)) return continuous<Where::RawHost>(fn, self, bounds, levels, p0, p1, inplace);
        check_host_table(self, levels, "levels");
        // `gelu` is the one operator the reference implements for host tensors, and there it returns a fresh tensor
        // (fewbit/cpu/gelu.cc:7-31); everything else honours the Tensor(a!) schema like the GPU side
        const bool write_back = inplace && fn != FEWBIT_GELU;
        if (write_back) {
            if (const Tensor base = whole_view_base(self); base.defined()) {
                continuous_with_node(true, fn, base, bounds, levels, p0, p1, true);
                return after_base_write(self, base);
            }
        }
        return continuous_with_node(true, fn, self, bounds, levels, p0, p1, write_back);
    } else if constexpr (W == Where::RawHost) {     // no autograd node: nothing to save, plain activation
        check_host_table(self, bounds, "bounds");      // the same argument errors with and without autograd
        check_host_table(self, levels, "levels");
        Tensor y = host_activation(fn, self, p0, p1);
        if (!inplace || fn == FEWBIT_GELU) return y;
        self.copy_(y);                                 // (copy_ bumps the version counter itself)
        return self;
    } else {
        // no autograd node (inference_mode): same kernel, the packed state goes to a scratch buffer and is dropped
        Tensor out = inplace ? self : torch::empty_like(self);
        launch_quantize(fn, self, out, bounds, p0, p1);
        if (inplace) note_inplace_write(self);
        return out;
    }
}

template <Where W> Tensor stepwise1(int fn, const Tensor &self, double p0 = 0.0, double p1 = 0.0, bool inplace = true) {
    if constexpr (W == Where::AutogradGpu) {
        if (!needs_node(self)) return stepwise1<Where::RawGpu>(fn, self, p0, p1, inplace);
        if (inplace) {
            if (const Tensor base = whole_view_base(self); base.defined()) {       // see whole_view_base
                stepwise1_with_node(false, fn, base, p0, p1, true);
                return after_base_write(self, base);
            }
        }
        return stepwise1_with_node(false, fn, self, p0, p1, inplace);
    } else if constexpr (W == Where::AutogradHost) {
        if (!needs_node(self)) return stepwise1<Where::RawHost>(fn, self, p0, p1, inplace);
        if (inplace) {
            if (const Tensor base = whole_view_base(self); base.defined()) {
                stepwise1_with_node(true, fn, base, p0, p1, true);
                return after_base_write(self, base);
            }
        }
        return stepwise1_with_node(true, fn, self, p0, p1, inplace);
    } else if constexpr (W == Where::RawHost) {
        Tensor y = host_step1_activation(fn, self, p0, p1);
        if (!inplace) return y;
        self.copy_(y);
        return self;
    } else {
        check_input(self, "self");
        Tensor out = inplace ? self : torch::empty_like(self);
        launch_step1(fn, self, out, p0, p1);
        if (inplace) note_inplace_write(self);
        return out;
    }
}

}  // namespace

// ---- op implementations (names follow the schema list below), one instantiation per dispatch key ----

template <Where W> Tensor hardshrink(const Tensor &self, double lambd) { return stepwise1<W>(FEWBIT_HARDSHRINK, self, lambd); }
template <Where W> Tensor hardsigmoid(const Tensor &self) { return stepwise1<W>(FEWBIT_HARDSIGMOID, self); }
template <Where W> Tensor hardtanh(const Tensor &self, double min_val, double max_val) { return stepwise1<W>(FEWBIT_HARDTANH, self, min_val, max_val); }
template <Where W> Tensor leaky_relu(const Tensor &self, double negative_slope) { return stepwise1<W>(FEWBIT_LEAKY_RELU, self, negative_slope); }
template <Where W> Tensor relu(const Tensor &self) { return stepwise1<W>(FEWBIT_RELU, self); }
template <Where W> Tensor relu6(const Tensor &self) { return stepwise1<W>(FEWBIT_RELU6, self); }
template <Where W> Tensor softshrink(const Tensor &self, double lambd) { return stepwise1<W>(FEWBIT_SOFTSHRINK, self, lambd); }
template <Where W> Tensor threshold(const Tensor &self, double threshold, double value) { return stepwise1<W>(FEWBIT_THRESHOLD, self, threshold, value); }

template <Where W> Tensor celu(const Tensor &self, const Tensor &b, const Tensor &l, double alpha) { return continuous<W>(FEWBIT_CELU, self, b, l, alpha); }
template <Where W> Tensor elu(const Tensor &self, const Tensor &b, const Tensor &l, double alpha) { return continuous<W>(FEWBIT_ELU, self, b, l, alpha); }
template <Where W> Tensor gelu(const Tensor &self, const Tensor &b, const Tensor &l) { return continuous<W>(FEWBIT_GELU, self, b, l); }
template <Where W> Tensor hardswish(const Tensor &self, const Tensor &b, const Tensor &l) { return continuous<W>(FEWBIT_HARDSWISH, self, b, l); }
template <Where W> Tensor logsigmoid(const Tensor &self, const Tensor &b, const Tensor &l) { return continuous<W>(FEWBIT_LOGSIGMOID, self, b, l); }
template <Where W> Tensor mish(const Tensor &self, const Tensor &b, const Tensor &l) { return continuous<W>(FEWBIT_MISH, self, b, l); }
template <Where W> Tensor selu(const Tensor &self, const Tensor &b, const Tensor &l) { return continuous<W>(FEWBIT_SELU, self, b, l); }
template <Where W> Tensor sigmoid(const Tensor &self, const Tensor &b, const Tensor &l) { return continuous<W>(FEWBIT_SIGMOID, self, b, l); }
template <Where W> Tensor silu(const Tensor &self, const Tensor &b, const Tensor &l) { return continuous<W>(FEWBIT_SILU, self, b, l); }
template <Where W> Tensor softplus(const Tensor &self, const Tensor &b, const Tensor &l, double beta, double threshold) {
    return continuous<W>(FEWBIT_SOFTPLUS, self, b, l, beta, threshold);
}
template <Where W> Tensor softsign(const Tensor &self, const Tensor &b, const Tensor &l) { return continuous<W>(FEWBIT_SOFTSIGN, self, b, l); }
template <Where W> Tensor tanh(const Tensor &self, const Tensor &b, const Tensor &l) { return continuous<W>(FEWBIT_TANH, self, b, l); }
template <Where W> Tensor tanhshrink(const Tensor &self, const Tensor &b, const Tensor &l) { return continuous<W>(FEWBIT_TANHSHRINK, self, b, l); }

// Custom table on the identity.  The reference declares this schema without any kernel (fewbit/fewbit.cc:37,
// NotImplementedError in fewbit/functional/activations.py:137-139; module fewbit/modules/activations.py:97-134:
// "parity: whether stepwise function is odd or even under shift transformation; shift: shift of the origin").
// Semantics defined here (EXPERIMENTS.md section 8), with (sx, sy) = shift and the table (b', l') given on the half line
// t = |x - sx| >= 0, as fewbit/approx.py:92-101 produces it for `parity=True, domain=(0, x_max)`:
//   even  g(sx+t) = g(sx-t):            code = #{b' < |x - sx|}, level = l'[code]   -- folded inside the kernel, so
//                                        k bits address 2^k half-line levels (twice the resolution of a plain table)
//   odd   g(sx+t)-sy = -(g(sx-t)-sy):   equal to the plain table  borders {sx-b'} u {sx} u {sx+b'},
//                                        levels {2sy-l'} u {l'}  -- mirrored here (fp32, rounded once to the tensor
//                                        dtype) and run through the plain kernels; the sign costs the one extra bit
template <Where W>
Tensor stepwise_folded_impl(const Tensor &self, const Tensor &b, const Tensor &l, bool even, double sx, double sy,
                            bool inplace) {
    if (even) return continuous<W>(FEWBIT_IDENTITY_FOLD, self, b, l, sx, 0.0, inplace);
    TORCH_CHECK(b.dim() == 1 && l.dim() == 1, "fewbit: `bounds` and `levels` must be one-dimensional");
    TORCH_CHECK(b.numel() + 1 == l.numel(), "fewbit: size of `bounds` should be lesser than size of `levels` by one, got ",
                b.numel(), " and ", l.numel());
    TORCH_CHECK(l.numel() <= 128, "fewbit: an odd-parity table mirrors to twice its size; at most 128 levels, got ", l.numel());
    const Tensor bf = b.to(torch::kFloat), lf = l.to(torch::kFloat);
    const Tensor centre = torch::full({1}, sx, bf.options());
    const Tensor full_b = torch::cat({torch::rsub(bf.flip(0), sx), centre, bf.add(sx)}).to(self.scalar_type());
    const Tensor full_l = torch::cat({torch::rsub(lf.flip(0), 2.0 * sy), lf}).to(self.scalar_type());
    return continuous<W>(FEWBIT_IDENTITY, self, full_b, full_l, 0.0, 0.0, inplace);
}

template <Where W> Tensor stepwise_folded(const Tensor &self, const Tensor &b, const Tensor &l, bool even, double sx, double sy) {
    return stepwise_folded_impl<W>(self, b, l, even, sx, sy, /*inplace=*/true);
}

template <Where W> Tensor stepwise_folded_out(const Tensor &self, const Tensor &b, const Tensor &l, bool even, double sx, double sy) {
    return stepwise_folded_impl<W>(self, b, l, even, sx, sy, /*inplace=*/false);
}

// the reference's schema: integer shift only (fewbit/fewbit.cc:37); `stepwise_folded` takes real shifts
template <Where W>
Tensor stepwise(const Tensor &self, const Tensor &b, const Tensor &l, std::optional<bool> parity,
                c10::OptionalArrayRef<int64_t> shift) {
    if (!parity.has_value()) {
        TORCH_CHECK(!shift.has_value(), "fewbit: stepwise `shift` needs a `parity`");
        return continuous<W>(FEWBIT_IDENTITY, self, b, l);
    }
    double sx = 0.0, sy = 0.0;
    if (shift.has_value()) {
        sx = static_cast<double>((*shift)[0]);
        sy = static_cast<double>((*shift)[1]);
    }
    return stepwise_folded_impl<W>(self, b, l, *parity, sx, sy, /*inplace=*/true);
}

// Out-of-place variants (additions, not in the reference): same kernels writing to a fresh tensor.  The python layer
// uses them when the input is a VIEW (e.g. the 3-D output of nn.Linear): an in-place op on a view makes autograd
// rebase the view's history (CopySlices), whose backward costs a zero-fill and four full-size copies per call --
// seen as +5 % step time on RoBERTa-base before this path existed.
template <Where W> Tensor continuous_out(const Tensor &self, const Tensor &b, const Tensor &l, int64_t fn, double p0, double p1) {
    TORCH_CHECK(fn >= 0 && fn < FEWBIT_CONTINUOUS_COUNT, "fewbit: unknown continuous function id ", fn);
    return continuous<W>(static_cast<int>(fn), self, b, l, p0, p1, /*inplace=*/false);
}

template <Where W> Tensor stepwise1_out(const Tensor &self, int64_t fn, double p0, double p1) {
    TORCH_CHECK(fn >= 0 && fn < FEWBIT_STEPWISE_COUNT, "fewbit: unknown stepwise function id ", fn);
    return stepwise1<W>(static_cast<int>(fn), self, p0, p1, /*inplace=*/false);
}

// raw pieces, fewbit/cpu/gelu.cc:7-45: quantize(x, bounds) -> (gelu(x), state); out of place like the reference
std::tuple<Tensor, Tensor> quantize(const Tensor &inputs, const Tensor &bounds) {
    const Tensor x = inputs.contiguous();
    Tensor outputs = torch::empty_like(x);
    Tensor state = launch_quantize(FEWBIT_GELU, x, outputs, bounds, 0.0, 0.0);
    return std::make_tuple(outputs, state);
}

Tensor quantize_backward(const Tensor &grads, const Tensor &buffer, const Tensor &levels) {
    return launch_dequantize(grads, buffer, levels);
}

// ... and for host tensors (the reference's own home for these two, fewbit/fewbit.cc:6-7)
std::tuple<Tensor, Tensor> quantize_host(const Tensor &inputs, const Tensor &bounds) {
    Tensor outputs;
    Tensor state = host_quantize(FEWBIT_GELU, inputs, outputs, bounds, 0.0, 0.0);
    return std::make_tuple(outputs, state);
}

Tensor quantize_backward_host(const Tensor &grads, const Tensor &buffer, const Tensor &levels) {
    check_host_table(grads, levels, "levels");
    const Tensor lv = levels.contiguous();
    TORCH_CHECK(lv.numel() >= 2 && lv.numel() <= 256, "fewbit: number of levels must be in [2, 256], got ", lv.numel());
    return host_unpack_mul(grads, buffer, lv, bitwidth_of(lv.numel()));
}

template <Where W> void register_activations(torch::Library &m) {
    m.impl("hardshrink", &hardshrink<W>);
    m.impl("hardsigmoid", &hardsigmoid<W>);
    m.impl("hardtanh", &hardtanh<W>);
    m.impl("leaky_relu", &leaky_relu<W>);
    m.impl("relu", &relu<W>);
    m.impl("relu6", &relu6<W>);
    m.impl("softshrink", &softshrink<W>);
    m.impl("threshold", &threshold<W>);

    m.impl("celu", &celu<W>);
    m.impl("elu", &elu<W>);
    m.impl("gelu", &gelu<W>);
    m.impl("hardswish", &hardswish<W>);
    m.impl("logsigmoid", &logsigmoid<W>);
    m.impl("mish", &mish<W>);
    m.impl("selu", &selu<W>);
    m.impl("sigmoid", &sigmoid<W>);
    m.impl("silu", &silu<W>);
    m.impl("softplus", &softplus<W>);
    m.impl("softsign", &softsign<W>);
    m.impl("tanh", &tanh<W>);
    m.impl("tanhshrink", &tanhshrink<W>);

    m.impl("stepwise", &stepwise<W>);
    m.impl("stepwise_folded", &stepwise_folded<W>);
    m.impl("stepwise_folded_out", &stepwise_folded_out<W>);
    m.impl("continuous_out", &continuous_out<W>);
    m.impl("stepwise1_out", &stepwise1_out<W>);
}

}  // namespace fewbit_amd

TORCH_LIBRARY(fewbit, m) {
    m.def("quantize(Tensor inputs, Tensor bounds) -> (Tensor, Tensor)");
    m.def("quantize_backward(Tensor grads, Tensor buffer, Tensor levels) -> Tensor");

    m.def("hardshrink (Tensor(a!) self, float lambd = 0.5) -> Tensor(a!)");
    m.def("hardsigmoid(Tensor(a!) self) -> Tensor(a!)");
    m.def("hardtanh   (Tensor(a!) self, float min_val = -1.0, float max_val = 1.0) -> Tensor(a!)");
    m.def("leaky_relu (Tensor(a!) self, float negative_slope = 0.01) -> Tensor(a!)");
    m.def("relu       (Tensor(a!) self) -> Tensor(a!)");
    m.def("relu6      (Tensor(a!) self) -> Tensor(a!)");
    m.def("softshrink (Tensor(a!) self, float lambd = 0.5) -> Tensor(a!)");
    m.def("threshold  (Tensor(a!) self, float threshold, float value) -> Tensor(a!)");

    m.def("celu      (Tensor(a!) self, Tensor bounds, Tensor levels, float alpha = 1.0) -> Tensor(a!)");
    m.def("elu       (Tensor(a!) self, Tensor bounds, Tensor levels, float alpha = 1.0) -> Tensor(a!)");
    m.def("gelu      (Tensor(a!) self, Tensor bounds, Tensor levels) -> Tensor(a!)");
    m.def("hardswish (Tensor(a!) self, Tensor bounds, Tensor levels) -> Tensor(a!)");
    m.def("logsigmoid(Tensor(a!) self, Tensor bounds, Tensor levels) -> Tensor(a!)");
    m.def("mish      (Tensor(a!) self, Tensor bounds, Tensor levels) -> Tensor(a!)");
    m.def("selu      (Tensor(a!) self, Tensor bounds, Tensor levels) -> Tensor(a!)");
    m.def("sigmoid   (Tensor(a!) self, Tensor bounds, Tensor levels) -> Tensor(a!)");
    m.def("silu      (Tensor(a!) self, Tensor bounds, Tensor levels) -> Tensor(a!)");
    m.def("softplus  (Tensor(a!) self, Tensor bounds, Tensor levels, float beta = 1.0, float threshold = 20.0) -> Tensor(a!)");
    m.def("softsign  (Tensor(a!) self, Tensor bounds, Tensor levels) -> Tensor(a!)");
    m.def("tanh      (Tensor(a!) self, Tensor bounds, Tensor levels) -> Tensor(a!)");
    m.def("tanhshrink(Tensor(a!) self, Tensor bounds, Tensor levels) -> Tensor(a!)");

    m.def("stepwise   (Tensor(a!) self, Tensor bounds, Tensor levels, bool? parity=None, int[2]? shift=None) -> Tensor(a!)");

    // additions of this implementation (not part of the reference's operator set)
    m.def("stepwise_folded(Tensor(a!) self, Tensor bounds, Tensor levels, bool even, float shift_x = 0.0, float shift_y = 0.0) -> Tensor(a!)");
    m.def("stepwise_folded_out(Tensor self, Tensor bounds, Tensor levels, bool even, float shift_x = 0.0, float shift_y = 0.0) -> Tensor");
    m.def("continuous_out(Tensor self, Tensor bounds, Tensor levels, int fn, float p0 = 0.0, float p1 = 0.0) -> Tensor");
    m.def("stepwise1_out(Tensor self, int fn, float p0 = 0.0, float p1 = 0.0) -> Tensor");
}

TORCH_LIBRARY_IMPL(fewbit, AutogradCUDA, m) { fewbit_amd::register_activations<fewbit_amd::Where::AutogradGpu>(m); }

TORCH_LIBRARY_IMPL(fewbit, CUDA, m) {
    fewbit_amd::register_activations<fewbit_amd::Where::RawGpu>(m);
    m.impl("quantize", fewbit_amd::quantize);
    m.impl("quantize_backward", fewbit_amd::quantize_backward);
}

TORCH_LIBRARY_IMPL(fewbit, AutogradCPU, m) { fewbit_amd::register_activations<fewbit_amd::Where::AutogradHost>(m); }

TORCH_LIBRARY_IMPL(fewbit, CPU, m) {
    fewbit_amd::register_activations<fewbit_amd::Where::RawHost>(m);
    m.impl("quantize", fewbit_amd::quantize_host);
    m.impl("quantize_backward", fewbit_amd::quantize_backward_host);
}

// Run-time switch of the autograd routes (header comment): name in {"direct_node", "base_dirty", "fresh_view"};
// value 0 / 1 sets, -1 only queries.  Returns the previous effective setting (0 / 1), -1 for an unknown name, -2 when the
// route cannot be enabled because the library was built without the internal-API code (FEWBIT_AUTOGRAD_INTERNALS = 0).
extern "C" __attribute__((visibility("default"))) int fewbit_torch_route(const char *name, int value) {
    namespace r = fewbit_amd::route;
    for (int w = 0; w < r::Count; ++w) {
        if (std::strcmp(name, r::kNames[w]) != 0) continue;
        const int prev = r::on(static_cast<r::Which>(w)) ? 1 : 0;
        if (value >= 0) {
            if (value > 0 && r::kNeedsInternals[w] && !FEWBIT_AUTOGRAD_INTERNALS) return -2;
            r::g_state[w].store(value > 0 ? 1 : 0, std::memory_order_relaxed);
        }
        return prev;
    }
    return -1;
}

extern "C" __attribute__((visibility("default"))) int fewbit_torch_autograd_internals(void) { return FEWBIT_AUTOGRAD_INTERNALS; }
