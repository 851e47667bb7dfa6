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
// There is no CPU implementation behind these ops: a CPU tensor raises from the dispatcher.
#include <torch/library.h>
#include <torch/torch.h>

// ROCm builds of PyTorch expose HIP devices as device type `cuda`; these are the matching guard / stream types
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

Tensor new_state(const Tensor &like, int64_t numel, int nbits) {
    const auto nbytes = static_cast<int64_t>(fewbit_hip_state_nbytes(static_cast<size_t>(numel), nbits));
    return torch::empty({nbytes}, torch::TensorOptions().device(like.device()).dtype(torch::kUInt8));
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

// ---- autograd Functions --------------------------------------------------------------------------

// all 13 continuous activations (+ custom `stepwise` tables): state and levels are what is saved
struct ContinuousFunction : public torch::autograd::Function<ContinuousFunction> {
    static Tensor forward(AutogradContext *ctx, Tensor self, const Tensor &bounds, const Tensor &levels, int64_t fn,
                          double p0, double p1, bool inplace) {
        TORCH_CHECK(bounds.numel() + 1 == levels.numel(),
                    "fewbit: size of `bounds` should be lesser than size of `levels` by one, got ", bounds.numel(),
                    " and ", levels.numel());
        check_table(self, levels, "levels");
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
        Tensor state = new_state(self, self.numel(), 1);
        c10::hip::HIPGuardMasqueradingAsCUDA guard(self.device());
        check_status(fewbit_hip_stepwise1_forward(static_cast<int>(fn), dtype_code(self), self.data_ptr(), out.data_ptr(),
                                                  state.data_ptr<uint8_t>(), static_cast<size_t>(self.numel()), p0, p1,
                                                  current_stream(self)),
                     "stepwise1_forward");
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
        Tensor gy = grad_output[0].contiguous();
        check_input(gy, "grad_output");
        Tensor gx = torch::empty_like(gy);
        c10::hip::HIPGuardMasqueradingAsCUDA guard(gy.device());
        check_status(fewbit_hip_stepwise1_backward(static_cast<int>(fn), dtype_code(gy), gy.data_ptr(),
                                                   saved[0].data_ptr<uint8_t>(), gx.data_ptr(),
                                                   static_cast<size_t>(gy.numel()), p0, current_stream(gy)),
                     "stepwise1_backward");
        return {gx, Tensor(), Tensor(), Tensor(), Tensor()};
    }
};

Tensor continuous(int fn, const Tensor &self, const Tensor &bounds, const Tensor &levels, double p0 = 0.0, double p1 = 0.0) {
    return ContinuousFunction::apply(self, bounds, levels, static_cast<int64_t>(fn), p0, p1, /*inplace=*/true);
}

Tensor stepwise1(int fn, const Tensor &self, double p0 = 0.0, double p1 = 0.0) {
    return Stepwise1Function::apply(self, static_cast<int64_t>(fn), p0, p1, /*inplace=*/true);
}

}  // namespace

// ---- op implementations (names follow the schema list below) ------------------------------------

Tensor hardshrink(const Tensor &self, double lambd) { return stepwise1(FEWBIT_HARDSHRINK, self, lambd); }
Tensor hardsigmoid(const Tensor &self) { return stepwise1(FEWBIT_HARDSIGMOID, self); }
Tensor hardtanh(const Tensor &self, double min_val, double max_val) { return stepwise1(FEWBIT_HARDTANH, self, min_val, max_val); }
Tensor leaky_relu(const Tensor &self, double negative_slope) { return stepwise1(FEWBIT_LEAKY_RELU, self, negative_slope); }
Tensor relu(const Tensor &self) { return stepwise1(FEWBIT_RELU, self); }
Tensor relu6(const Tensor &self) { return stepwise1(FEWBIT_RELU6, self); }
Tensor softshrink(const Tensor &self, double lambd) { return stepwise1(FEWBIT_SOFTSHRINK, self, lambd); }
Tensor threshold(const Tensor &self, double threshold, double value) { return stepwise1(FEWBIT_THRESHOLD, self, threshold, value); }

Tensor celu(const Tensor &self, const Tensor &b, const Tensor &l, double alpha) { return continuous(FEWBIT_CELU, self, b, l, alpha); }
Tensor elu(const Tensor &self, const Tensor &b, const Tensor &l, double alpha) { return continuous(FEWBIT_ELU, self, b, l, alpha); }
Tensor gelu(const Tensor &self, const Tensor &b, const Tensor &l) { return continuous(FEWBIT_GELU, self, b, l); }
Tensor hardswish(const Tensor &self, const Tensor &b, const Tensor &l) { return continuous(FEWBIT_HARDSWISH, self, b, l); }
Tensor logsigmoid(const Tensor &self, const Tensor &b, const Tensor &l) { return continuous(FEWBIT_LOGSIGMOID, self, b, l); }
Tensor mish(const Tensor &self, const Tensor &b, const Tensor &l) { return continuous(FEWBIT_MISH, self, b, l); }
Tensor selu(const Tensor &self, const Tensor &b, const Tensor &l) { return continuous(FEWBIT_SELU, self, b, l); }
Tensor sigmoid(const Tensor &self, const Tensor &b, const Tensor &l) { return continuous(FEWBIT_SIGMOID, self, b, l); }
Tensor silu(const Tensor &self, const Tensor &b, const Tensor &l) { return continuous(FEWBIT_SILU, self, b, l); }
Tensor softplus(const Tensor &self, const Tensor &b, const Tensor &l, double beta, double threshold) {
    return continuous(FEWBIT_SOFTPLUS, self, b, l, beta, threshold);
}
Tensor softsign(const Tensor &self, const Tensor &b, const Tensor &l) { return continuous(FEWBIT_SOFTSIGN, self, b, l); }
Tensor tanh(const Tensor &self, const Tensor &b, const Tensor &l) { return continuous(FEWBIT_TANH, self, b, l); }
Tensor tanhshrink(const Tensor &self, const Tensor &b, const Tensor &l) { return continuous(FEWBIT_TANHSHRINK, self, b, l); }

// Custom table on the identity.  The reference declares this schema without any kernel (fewbit/fewbit.cc:37,
// NotImplementedError in fewbit/functional/activations.py:137-139; module fewbit/modules/activations.py:97-134:
// "parity: whether stepwise function is odd or even under shift transformation; shift: shift of the origin").
// Semantics defined here, with (sx, sy) = shift and the table (b', l') given on the half line
// t = |x - sx| >= 0, as fewbit/approx.py:92-101 produces it for `parity=True, domain=(0, x_max)`:
//   even  g(sx+t) = g(sx-t):            code = #{b' < |x - sx|}, level = l'[code]   -- folded inside the kernel, so
//                                        k bits address 2^k half-line levels (twice the resolution of a plain table)
//   odd   g(sx+t)-sy = -(g(sx-t)-sy):   equal to the plain table  borders {sx-b'} u {sx} u {sx+b'},
//                                        levels {2sy-l'} u {l'}  -- mirrored here (fp32, rounded once to the tensor
//                                        dtype) and run through the plain kernels; the sign costs the one extra bit
Tensor stepwise_folded_impl(const Tensor &self, const Tensor &b, const Tensor &l, bool even, double sx, double sy,
                            bool inplace) {
    if (even) return ContinuousFunction::apply(self, b, l, FEWBIT_IDENTITY_FOLD, sx, 0.0, inplace);
    TORCH_CHECK(b.dim() == 1 && l.dim() == 1, "fewbit: `bounds` and `levels` must be one-dimensional");
    TORCH_CHECK(b.numel() + 1 == l.numel(), "fewbit: size of `bounds` should be lesser than size of `levels` by one, got ",
                b.numel(), " and ", l.numel());
    TORCH_CHECK(l.numel() <= 128, "fewbit: an odd-parity table mirrors to twice its size; at most 128 levels, got ", l.numel());
    const Tensor bf = b.to(torch::kFloat), lf = l.to(torch::kFloat);
    const Tensor centre = torch::full({1}, sx, bf.options());
    const Tensor full_b = torch::cat({torch::rsub(bf.flip(0), sx), centre, bf.add(sx)}).to(self.scalar_type());
    const Tensor full_l = torch::cat({torch::rsub(lf.flip(0), 2.0 * sy), lf}).to(self.scalar_type());
    return ContinuousFunction::apply(self, full_b, full_l, FEWBIT_IDENTITY, 0.0, 0.0, inplace);
}

Tensor stepwise_folded(const Tensor &self, const Tensor &b, const Tensor &l, bool even, double sx, double sy) {
    return stepwise_folded_impl(self, b, l, even, sx, sy, /*inplace=*/true);
}

Tensor stepwise_folded_out(const Tensor &self, const Tensor &b, const Tensor &l, bool even, double sx, double sy) {
    return stepwise_folded_impl(self, b, l, even, sx, sy, /*inplace=*/false);
}

// the reference's schema: integer shift only (fewbit/fewbit.cc:37); `stepwise_folded` takes real shifts
Tensor stepwise(const Tensor &self, const Tensor &b, const Tensor &l, std::optional<bool> parity,
                c10::OptionalArrayRef<int64_t> shift) {
    if (!parity.has_value()) {
        TORCH_CHECK(!shift.has_value(), "fewbit: stepwise `shift` needs a `parity`");
        return continuous(FEWBIT_IDENTITY, self, b, l);
    }
    double sx = 0.0, sy = 0.0;
    if (shift.has_value()) {
        sx = static_cast<double>((*shift)[0]);
        sy = static_cast<double>((*shift)[1]);
    }
    return stepwise_folded_impl(self, b, l, *parity, sx, sy, /*inplace=*/true);
}

// Out-of-place variants (additions, not in the reference): same kernels writing to a fresh tensor.  The python layer
// uses them when the input is a VIEW (e.g. the 3-D output of nn.Linear): an in-place op on a view makes autograd
// rebase the view's history (CopySlices), whose backward costs a zero-fill and four full-size copies per call --
// seen as +5 % step time on RoBERTa-base before this path existed.
Tensor continuous_out(const Tensor &self, const Tensor &b, const Tensor &l, int64_t fn, double p0, double p1) {
    TORCH_CHECK(fn >= 0 && fn < FEWBIT_CONTINUOUS_COUNT, "fewbit: unknown continuous function id ", fn);
    return ContinuousFunction::apply(self, b, l, fn, p0, p1, /*inplace=*/false);
}

Tensor stepwise1_out(const Tensor &self, int64_t fn, double p0, double p1) {
    TORCH_CHECK(fn >= 0 && fn < FEWBIT_STEPWISE_COUNT, "fewbit: unknown stepwise function id ", fn);
    return Stepwise1Function::apply(self, fn, p0, p1, /*inplace=*/false);
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

TORCH_LIBRARY_IMPL(fewbit, AutogradCUDA, m) {
    m.impl("hardshrink", fewbit_amd::hardshrink);
    m.impl("hardsigmoid", fewbit_amd::hardsigmoid);
    m.impl("hardtanh", fewbit_amd::hardtanh);
    m.impl("leaky_relu", fewbit_amd::leaky_relu);
    m.impl("relu", fewbit_amd::relu);
    m.impl("relu6", fewbit_amd::relu6);
    m.impl("softshrink", fewbit_amd::softshrink);
    m.impl("threshold", fewbit_amd::threshold);

    m.impl("celu", fewbit_amd::celu);
    m.impl("elu", fewbit_amd::elu);
    m.impl("gelu", fewbit_amd::gelu);
    m.impl("hardswish", fewbit_amd::hardswish);
    m.impl("logsigmoid", fewbit_amd::logsigmoid);
    m.impl("mish", fewbit_amd::mish);
    m.impl("selu", fewbit_amd::selu);
    m.impl("sigmoid", fewbit_amd::sigmoid);
    m.impl("silu", fewbit_amd::silu);
    m.impl("softplus", fewbit_amd::softplus);
    m.impl("softsign", fewbit_amd::softsign);
    m.impl("tanh", fewbit_amd::tanh);
    m.impl("tanhshrink", fewbit_amd::tanhshrink);

    m.impl("stepwise", fewbit_amd::stepwise);
    m.impl("stepwise_folded", fewbit_amd::stepwise_folded);
    m.impl("stepwise_folded_out", fewbit_amd::stepwise_folded_out);
    m.impl("continuous_out", fewbit_amd::continuous_out);
    m.impl("stepwise1_out", fewbit_amd::stepwise1_out);
}

TORCH_LIBRARY_IMPL(fewbit, CUDA, m) {
    m.impl("quantize", fewbit_amd::quantize);
    m.impl("quantize_backward", fewbit_amd::quantize_backward);
}
