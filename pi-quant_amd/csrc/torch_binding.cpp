// Native front end of piquant.torch.quantize / dequantize for ROCm tensors.
//
// The reference's tensor API (python/src/piquant/torch.py:70-129) is a few lines of Python around one FFI call.  On a GPU
// the kernel for a 10^6-element tensor takes ~3 us, so those few lines (argument checks, torch.empty, current stream,
// ctypes marshalling of nine arguments) were most of a call: 8.8 us per piquant.torch.quantize against 5.2 us for PyTorch's
// own device quantize_per_tensor.  This extension does the whole call in C++ -- checks, at::empty on the tensor's device,
// the current HIP stream from c10, then the SAME C ABI entry points of libpiquant.so (nothing is computed here) -- and is used
// by piquant/torch.py whenever it is present and the tensor lives on the device.  Without it the ctypes path is taken; both
// end in piquant_quantize / piquant_dequantize, so results are identical by construction (and tested).
//
// Built in-tree by build_torch_binding.py (g++, no hipcc: host code only) into piquant/_piquant_torch.so.
#include <torch/extension.h>
#include <torch/csrc/Dtype.h>
#include <torch/csrc/Exceptions.h>
#include <torch/csrc/autograd/python_variable.h>

#include <ATen/hip/EmptyTensor.h>
#include <ATen/quantized/Quantizer.h>
#include <c10/hip/HIPStream.h>

#include "piquant.h"
#include "piquant_hip.h"

namespace {

piquant_dtype_t code_of(at::ScalarType t) {
    switch (t) {
        case at::kFloat: return PIQUANT_DTYPE_F32;
        case at::kBFloat16: return PIQUANT_DTYPE_BF16;
        case at::kByte:
        case at::kQUInt8: return PIQUANT_DTYPE_UINT8;
        case at::kQUInt4x2: return PIQUANT_DTYPE_UINT4;
        case at::kQUInt2x4: return PIQUANT_DTYPE_UINT2;
        default: TORCH_CHECK(false, "Unsupported quant_dtype: ", t, " (no piquant counterpart)");
    }
}

bool is_float_type(at::ScalarType t) { return t == at::kFloat || t == at::kBFloat16; }

int64_t packed_nbytes(int64_t numel, piquant_dtype_t q) {
    return q == PIQUANT_DTYPE_UINT8 ? numel : (q == PIQUANT_DTYPE_UINT4 ? (numel + 1) / 2 : (numel + 3) / 4);
}

// stream-ordered on the tensor's current stream, device pointers known: what piquant.torch._ctx_for sets up
// torch.empty(like.shape, dtype=dtype, device=like.device) (reference torch.py:87,117, plus the device) without the dispatcher: at::empty spends
// 0.7 us of a 4.9 us call finding the kernel that ends in the two functions called here, and builds a fresh "unknown quantizer" object per
// quantized tensor -- one per dtype is kept instead (it is stateless: such a tensor has no scale of its own, its bytes are read with
// piquant.torch.packed_bytes or handed back to dequantize).
at::Tensor fresh_tensor(const at::Tensor& like, at::ScalarType dtype) {
    if (c10::isQIntType(dtype)) {
        static const at::QuantizerPtr unknown[3] = {at::make_unknown_quantizer(at::kQUInt8), at::make_unknown_quantizer(at::kQUInt4x2),
                                                    at::make_unknown_quantizer(at::kQUInt2x4)};
        const int i = dtype == at::kQUInt8 ? 0 : (dtype == at::kQUInt4x2 ? 1 : 2);
        return at::new_qtensor(like.sizes(), like.options().dtype(dtype), unknown[i]);
    }
    return at::Tensor(at::detail::empty_cuda(like.sizes(), dtype, like.device(), c10::nullopt));
}

piquant_context_t* prepare(int64_t handle, const at::Tensor& t) {
    auto* ctx = reinterpret_cast<piquant_context_t*>(static_cast<intptr_t>(handle));
    TORCH_CHECK(ctx != nullptr, "piquant context handle is NULL");
    TORCH_CHECK(piquant_hip_device(ctx) == t.get_device(), "context is bound to device ", piquant_hip_device(ctx), " but the tensor lives on device ",
                t.get_device());
    piquant_hip_set_stream(ctx, c10::hip::getCurrentHIPStream(t.get_device()).stream());
    piquant_hip_set_blocking(ctx, 0);
    piquant_hip_assume_device_pointers(ctx, 1);
    return ctx;
}

at::Tensor quantize(int64_t handle, const at::Tensor& tensor, double scale, int64_t zero_point, at::ScalarType dtype, int64_t round_mode,
                    const c10::optional<at::Tensor>& out_opt, bool uniform) {
    TORCH_CHECK(tensor.is_cuda(), "the native path takes device tensors");
    TORCH_CHECK(is_float_type(tensor.scalar_type()), "quantize needs a float32 or bfloat16 tensor, got ", tensor.scalar_type());
    const piquant_dtype_t dt_out = code_of(dtype);
    TORCH_CHECK(!is_float_type(dtype), "Unsupported quantized dtype: ", dtype, " (dtype= must be uint8 / quint8, quint4x2 or quint2x4)");
    const at::Tensor x = tensor.is_contiguous() ? tensor : tensor.contiguous();
    at::Tensor out;
    if (out_opt.has_value()) {
        out = *out_opt;
        TORCH_CHECK(out.is_contiguous() && out.device() == x.device(), "out= must be a contiguous tensor on the input's device");
        // the TENSOR must hold the result, not merely the storage it views: a short slice of a large buffer is too small
        if (out.scalar_type() == at::kByte) {
            const int64_t need = static_cast<int64_t>(packed_nbytes(x.numel(), dt_out));
            TORCH_CHECK(out.numel() >= need, "out= holds ", out.numel(), " bytes, ", need, " are needed");
        } else {
            const bool fits = !is_float_type(out.scalar_type()) && code_of(out.scalar_type()) == dt_out && out.numel() == x.numel();
            TORCH_CHECK(fits, "out= must be a quantized tensor of the requested dtype with the input's number of elements (or a uint8 buffer of the packed bytes)");
        }
    } else {
        out = fresh_tensor(x, dtype);
    }
    piquant_context_t* ctx = prepare(handle, x);
    (uniform ? piquant_hip_quantize_uniform : piquant_quantize)(ctx, x.data_ptr(), code_of(x.scalar_type()), out.data_ptr(), dt_out, static_cast<size_t>(x.numel()),
                                                                static_cast<float>(scale), zero_point, static_cast<piquant_round_mode_t>(round_mode));
    return out;
}

// `tensor` is a quantized tensor (quint8 / quint4x2 / quint2x4 / uint8); its own dtype and shape describe it
at::Tensor dequantize(int64_t handle, const at::Tensor& tensor, double scale, int64_t zero_point, at::ScalarType dtype, int64_t reduce_op,
                      const c10::optional<at::Tensor>& out_opt, bool uniform) {
    TORCH_CHECK(tensor.is_cuda(), "the native path takes device tensors");
    TORCH_CHECK(is_float_type(dtype), "Unsupported dequantized dtype: ", dtype, " (dtype= must be float32 or bfloat16)");
    TORCH_CHECK(!is_float_type(tensor.scalar_type()), "Unsupported quantized dtype: ", tensor.scalar_type(),
                " (dequantize needs a uint8 / quint8, quint4x2 or quint2x4 tensor)");
    const at::Tensor q = tensor.is_contiguous() ? tensor : tensor.contiguous();
    at::Tensor out;
    if (out_opt.has_value()) {
        out = *out_opt;
        TORCH_CHECK(out.scalar_type() == dtype && out.is_contiguous() && out.device() == q.device() && out.numel() == q.numel(),
                    "out= must be a contiguous tensor of the requested dtype, the input's device and the input's number of elements");
    } else {
        TORCH_CHECK(reduce_op == PIQUANT_REDUCE_OP_SET, "reduce_op='add' accumulates into out=; pass the accumulator tensor");
        out = fresh_tensor(q, dtype);
    }
    piquant_context_t* ctx = prepare(handle, q);
    (uniform ? piquant_hip_dequantize_uniform : piquant_dequantize)(ctx, q.data_ptr(), code_of(q.scalar_type()), out.data_ptr(), code_of(dtype), static_cast<size_t>(q.numel()),
                                                                    static_cast<float>(scale), zero_point, static_cast<piquant_reduce_op_t>(reduce_op));
    return out;
}

// The calling thread's default context of a device, as piquant.Context.get() hands it out: resolved through a Python callback once per (thread,
// device) and remembered here, so that a call without ctx= crosses into Python for nothing (the dictionary look-ups, the device object and the three
// attribute stores of the Python-side resolution were 0.4 us of a 5 us call).
py::object& resolver() {
    static py::object* r = new py::object();   // leaked on purpose: destroyed after the interpreter otherwise
    return *r;
}
constexpr int kMaxDevices = 64;
thread_local int64_t tl_default_handle[kMaxDevices] = {};

int64_t default_handle(const at::Tensor& t) {
    const int dev = static_cast<int>(t.get_device());
    TORCH_CHECK(dev >= 0 && dev < kMaxDevices, "device index ", dev, " out of range");
    if (tl_default_handle[dev] == 0) {
        TORCH_CHECK(!resolver().is_none() && resolver().ptr() != nullptr, "piquant.torch has not registered its default-context resolver");
        tl_default_handle[dev] = resolver()(dev).cast<int64_t>();
    }
    return tl_default_handle[dev];
}

[[noreturn]] void raise_assertion(const std::string& msg) {
    PyErr_SetString(PyExc_AssertionError, msg.c_str());
    throw py::error_already_set();
}

int64_t round_mode_code(const std::string& name) {
    if (name == "nearest") return PIQUANT_NEAREST;
    if (name == "stochastic") return PIQUANT_STOCHASTIC;
    throw py::key_error(name);
}

int64_t reduce_op_code(const std::string& name) {
    if (name == "set") return PIQUANT_REDUCE_OP_SET;
    if (name == "add") return PIQUANT_REDUCE_OP_ADD;
    throw py::key_error(name);
}

bool is_quant_type(at::ScalarType t) { return t == at::kByte || t == at::kQUInt8 || t == at::kQUInt4x2 || t == at::kQUInt2x4; }

// piquant.torch.quantize / dequantize for a device tensor and no ctx=: everything but the Python function call itself happens here
at::Tensor quantize_default(const at::Tensor& tensor, double scale, int64_t zero_point, at::ScalarType dtype, const std::string& round_mode,
                            const c10::optional<at::Tensor>& out, bool uniform) {
    if (!is_quant_type(dtype)) raise_assertion("Unsupported quantized dtype: " + std::string(c10::toString(dtype)));
    return quantize(default_handle(tensor), tensor, scale, zero_point, dtype, round_mode_code(round_mode), out, uniform);
}

at::Tensor dequantize_default(const at::Tensor& tensor, double scale, int64_t zero_point, at::ScalarType dtype, const std::string& reduce_op,
                              const c10::optional<at::Tensor>& out, bool uniform) {
    const int64_t op = reduce_op_code(reduce_op);
    if (!out.has_value() && op == PIQUANT_REDUCE_OP_ADD) throw py::value_error("reduce_op='add' accumulates into out=; pass the accumulator tensor");
    return dequantize(default_handle(tensor), tensor, scale, zero_point, dtype, op, out, uniform);
}

// piquant.torch.quantize / dequantize THEMSELVES when this module is present.  A call costs ~4 us of HIP launch whatever is done, so what the front end
// adds is what can be won or lost against PyTorch's own device quantize_per_tensor: a Python wrapper frame is 0.25 us, pybind11's keyword handling
// 0.5 us (measured: it made the call SLOWER than the Python wrapper around a positional pybind11 function), so the two entry points are plain CPython
// vectorcall functions: keyword names are matched by pointer (call-site identifiers are interned), the common case -- device tensor, default context,
// Python numbers, a torch.dtype -- is served right here, and EVERYTHING else (host tensors, ctx=, raw packed buffers, tensors for scale, misspelt
// keywords, unknown mode names) is handed, arguments untouched, to the Python implementation registered by piquant/torch.py, which is also what runs
// when this module is not built.  Same results and same exceptions either way.
py::object& python_impl(int which) {
    static py::object* f = new py::object[2];   // leaked on purpose (see resolver())
    return f[which];
}

struct Interned {
    PyObject *scale, *zero_point, *dtype, *round_mode, *reduce_op, *ctx, *out, *uniform, *quant_dtype, *shape, *nearest, *stochastic, *set, *add;
};
const Interned& names() {
    static const Interned n = {PyUnicode_InternFromString("scale"),      PyUnicode_InternFromString("zero_point"), PyUnicode_InternFromString("dtype"),
                               PyUnicode_InternFromString("round_mode"), PyUnicode_InternFromString("reduce_op"),  PyUnicode_InternFromString("ctx"),
                               PyUnicode_InternFromString("out"),        PyUnicode_InternFromString("uniform"),    PyUnicode_InternFromString("quant_dtype"),
                               PyUnicode_InternFromString("shape"),      PyUnicode_InternFromString("nearest"),    PyUnicode_InternFromString("stochastic"),
                               PyUnicode_InternFromString("set"),        PyUnicode_InternFromString("add")};
    return n;
}

bool plain_number(PyObject* o) { return PyFloat_CheckExact(o) || (PyLong_CheckExact(o)); }
bool absent(PyObject* o) { return o == nullptr || o == Py_None; }

// -1: not one of the two names (the Python implementation raises what the reference raises for it)
int choice(PyObject* o, PyObject* first, PyObject* second, const char* first_s, const char* second_s) {
    if (o == nullptr || o == first) return 0;
    if (o == second) return 1;
    if (!PyUnicode_CheckExact(o)) return -1;
    if (PyUnicode_CompareWithASCIIString(o, first_s) == 0) return 0;
    if (PyUnicode_CompareWithASCIIString(o, second_s) == 0) return 1;
    return -1;
}

PyObject* raise_current_exception() {
    try {
        throw;
    } catch (py::error_already_set& e) {
        e.restore();
    } catch (py::builtin_exception& e) {
        e.set_error();
    } catch (const std::exception&) {
        torch::translate_exception_to_python(std::current_exception());
    }
    return nullptr;
}

PyObject* quantize_entry(PyObject*, PyObject* const* args, Py_ssize_t nargs, PyObject* kwnames) {
    const Interned& n = names();
    PyObject *scale = nullptr, *zero_point = nullptr, *dtype = nullptr, *round_mode = nullptr, *ctx = nullptr, *out = nullptr, *uniform = nullptr;
    bool common = nargs == 1;
    const Py_ssize_t n_kw = kwnames != nullptr ? PyTuple_GET_SIZE(kwnames) : 0;
    for (Py_ssize_t i = 0; i < n_kw && common; ++i) {
        PyObject* const name = PyTuple_GET_ITEM(kwnames, i);
        PyObject* const v = args[nargs + i];
        if (name == n.scale) scale = v;
        else if (name == n.zero_point) zero_point = v;
        else if (name == n.dtype) dtype = v;
        else if (name == n.round_mode) round_mode = v;
        else if (name == n.ctx) ctx = v;
        else if (name == n.out) out = v;
        else if (name == n.uniform) uniform = v;
        else common = false;
    }
    common = common && scale != nullptr && zero_point != nullptr && dtype != nullptr && absent(ctx) && THPVariable_Check(args[0]) && THPDtype_Check(dtype) &&
             plain_number(scale) && PyLong_CheckExact(zero_point) && (absent(out) || THPVariable_Check(out)) && (uniform == nullptr || PyBool_Check(uniform));
    const int mode = common ? choice(round_mode, n.nearest, n.stochastic, "nearest", "stochastic") : -1;
    if (mode >= 0) {
        const at::Tensor& tensor = THPVariable_Unpack(args[0]);
        int overflow = 0;
        const long long zp = PyLong_AsLongLongAndOverflow(zero_point, &overflow);
        if (tensor.is_cuda() && is_float_type(tensor.scalar_type()) && overflow == 0) {
            try {
                const at::ScalarType dt = reinterpret_cast<THPDtype*>(dtype)->scalar_type;
                if (!is_quant_type(dt)) raise_assertion("Unsupported quantized dtype: " + std::string(c10::toString(dt)));
                const double s = PyFloat_CheckExact(scale) ? PyFloat_AS_DOUBLE(scale) : PyLong_AsDouble(scale);
                if (s == -1.0 && PyErr_Occurred()) return nullptr;
                c10::optional<at::Tensor> out_opt;
                if (!absent(out)) out_opt = THPVariable_Unpack(out);
                at::Tensor r = quantize(default_handle(tensor), tensor, s, zp, dt, mode == 0 ? PIQUANT_NEAREST : PIQUANT_STOCHASTIC, out_opt, uniform == Py_True);
                if (!absent(out)) {
                    Py_INCREF(out);
                    return out;
                }
                return THPVariable_Wrap(std::move(r));
            } catch (...) {
                return raise_current_exception();
            }
        }
    }
    return PyObject_Vectorcall(python_impl(0).ptr(), args, static_cast<size_t>(nargs), kwnames);
}

PyObject* dequantize_entry(PyObject*, PyObject* const* args, Py_ssize_t nargs, PyObject* kwnames) {
    const Interned& n = names();
    PyObject *scale = nullptr, *zero_point = nullptr, *dtype = nullptr, *reduce_op = nullptr, *ctx = nullptr, *out = nullptr, *uniform = nullptr, *quant_dtype = nullptr,
             *shape = nullptr;
    bool common = nargs == 1;
    const Py_ssize_t n_kw = kwnames != nullptr ? PyTuple_GET_SIZE(kwnames) : 0;
    for (Py_ssize_t i = 0; i < n_kw && common; ++i) {
        PyObject* const name = PyTuple_GET_ITEM(kwnames, i);
        PyObject* const v = args[nargs + i];
        if (name == n.scale) scale = v;
        else if (name == n.zero_point) zero_point = v;
        else if (name == n.dtype) dtype = v;
        else if (name == n.reduce_op) reduce_op = v;
        else if (name == n.ctx) ctx = v;
        else if (name == n.out) out = v;
        else if (name == n.uniform) uniform = v;
        else if (name == n.quant_dtype) quant_dtype = v;
        else if (name == n.shape) shape = v;
        else common = false;
    }
    common = common && scale != nullptr && zero_point != nullptr && dtype != nullptr && absent(ctx) && absent(quant_dtype) && absent(shape) &&
             THPVariable_Check(args[0]) && THPDtype_Check(dtype) && plain_number(scale) && PyLong_CheckExact(zero_point) && (absent(out) || THPVariable_Check(out)) &&
             (uniform == nullptr || PyBool_Check(uniform));
    const int op = common ? choice(reduce_op, n.set, n.add, "set", "add") : -1;
    if (op >= 0) {
        const at::Tensor& tensor = THPVariable_Unpack(args[0]);
        int overflow = 0;
        const long long zp = PyLong_AsLongLongAndOverflow(zero_point, &overflow);
        const at::ScalarType dt = reinterpret_cast<THPDtype*>(dtype)->scalar_type;
        // an accumulating call without out= and a non-float dtype= are the Python implementation's to refuse (ValueError, as the reference)
        if (tensor.is_cuda() && is_quant_type(tensor.scalar_type()) && overflow == 0 && is_float_type(dt) && !(op == 1 && absent(out))) {
            try {
                const double s = PyFloat_CheckExact(scale) ? PyFloat_AS_DOUBLE(scale) : PyLong_AsDouble(scale);
                if (s == -1.0 && PyErr_Occurred()) return nullptr;
                c10::optional<at::Tensor> out_opt;
                if (!absent(out)) out_opt = THPVariable_Unpack(out);
                at::Tensor r = dequantize(default_handle(tensor), tensor, s, zp, dt, op == 0 ? PIQUANT_REDUCE_OP_SET : PIQUANT_REDUCE_OP_ADD, out_opt, uniform == Py_True);
                if (!absent(out)) {
                    Py_INCREF(out);
                    return out;
                }
                return THPVariable_Wrap(std::move(r));
            } catch (...) {
                return raise_current_exception();
            }
        }
    }
    return PyObject_Vectorcall(python_impl(1).ptr(), args, static_cast<size_t>(nargs), kwnames);
}

PyMethodDef entry_points[] = {
    {"quantize", reinterpret_cast<PyCFunction>(reinterpret_cast<void (*)()>(quantize_entry)), METH_FASTCALL | METH_KEYWORDS,
     "quantize(tensor, *, scale, zero_point, dtype, round_mode='nearest', ctx=None, out=None, uniform=False)\n\n"
     "piquant.torch.quantize (reference python/src/piquant/torch.py:70-99); documented in piquant/torch.py"},
    {"dequantize", reinterpret_cast<PyCFunction>(reinterpret_cast<void (*)()>(dequantize_entry)), METH_FASTCALL | METH_KEYWORDS,
     "dequantize(tensor, *, scale, zero_point, dtype, reduce_op='set', ctx=None, out=None, quant_dtype=None, shape=None, uniform=False)\n\n"
     "piquant.torch.dequantize (reference python/src/piquant/torch.py:102-129); documented in piquant/torch.py"},
};

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.doc() = "native front end of piquant.torch for ROCm tensors (forwards to the C ABI of libpiquant.so)";
    m.def("quantize", &quantize, py::arg("handle"), py::arg("tensor"), py::arg("scale"), py::arg("zero_point"), py::arg("dtype"), py::arg("round_mode"),
          py::arg("out") = py::none(), py::arg("uniform") = false);
    m.def("dequantize", &dequantize, py::arg("handle"), py::arg("tensor"), py::arg("scale"), py::arg("zero_point"), py::arg("dtype"), py::arg("reduce_op"),
          py::arg("out") = py::none(), py::arg("uniform") = false);
    m.def("set_default_resolver", [](py::object f) { resolver() = std::move(f); }, "callable(device index) -> native handle of the calling thread's default context");
    m.def("forget_default_handles", [] { for (auto& h : tl_default_handle) h = 0; }, "drop the calling thread's remembered default contexts");
    m.def("set_python_implementations", [](py::object q, py::object dq) { python_impl(0) = std::move(q); python_impl(1) = std::move(dq); });
    m.add_object("quantize_entry", py::reinterpret_steal<py::object>(PyCFunction_NewEx(&entry_points[0], nullptr, nullptr)));
    m.add_object("dequantize_entry", py::reinterpret_steal<py::object>(PyCFunction_NewEx(&entry_points[1], nullptr, nullptr)));
    m.def("quantize_default", &quantize_default, py::arg("tensor"), py::arg("scale"), py::arg("zero_point"), py::arg("dtype"), py::arg("round_mode"),
          py::arg("out") = py::none(), py::arg("uniform") = false);
    m.def("dequantize_default", &dequantize_default, py::arg("tensor"), py::arg("scale"), py::arg("zero_point"), py::arg("dtype"), py::arg("reduce_op"),
          py::arg("out") = py::none(), py::arg("uniform") = false);
}
