// pybind_shim.cpp -- the compiled module `noLZSS._noLZSS` over the C ABI of libnolzss_hip.so.
//
// This is the binding a maintainer of the reference would put in place of the bodies of
// /root/reference/src/cpp/bindings.cpp for the factorize path: same module
// name, function names, argument names and defaults, result shapes and exception types; the compute
// goes to the HIP library through include/nolzss_hip.h and nothing else.  Host-only C++ (g++), no HIP
// or torch types.  Functions of the reference module that are not bound here are supplied by the
// ctypes mirror (nolzss_amd._noLZSS) when the package assembles its namespace (noLZSS/__init__.py).
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/nolzss_hip.h"

namespace py = pybind11;

namespace {

int g_device = 0;

// same exception types pybind11 gives the reference's C++ exceptions (bindings.cpp relies on the
// default translation: std::invalid_argument -> ValueError, std::runtime_error -> RuntimeError)
[[noreturn]] void raise_last(int rc) {
    const std::string msg = nolzss_last_error();
    if (rc == NOLZSS_ERR_INVALID_ARGUMENT) throw std::invalid_argument(msg);
    if (rc == NOLZSS_ERR_NOMEM) throw std::bad_alloc();
    throw std::runtime_error(msg);
}

struct Bytes {
    const uint8_t *ptr;
    size_t n;
};

// bindings.cpp:58-67: any 1-D buffer of 1-byte items, borrowed for the call
Bytes view(const py::buffer &b, const char *fn) {
    py::buffer_info info = b.request();
    if (info.itemsize != 1) throw std::invalid_argument(std::string(fn) + ": buffer must be a bytes-like object with itemsize==1");
    if (info.ndim != 1) throw std::invalid_argument(std::string(fn) + ": buffer must be a 1-dimensional bytes-like object");
    return Bytes{static_cast<const uint8_t *>(info.ptr), static_cast<size_t>(info.size)};
}

struct Owned {  // library-owned factor array
    nolzss_factor *f = nullptr;
    size_t z = 0;
    ~Owned() { nolzss_free(f); }
};

// The list of int tuples is what a Python caller waits for (5*10^7 of them for the 2^30-base benchmark text):
// built with the C API directly -- py::make_tuple goes through three py::object casts and a std::array per
// tuple, twice the time.
inline PyObject *py_u64(uint64_t v) {
    PyObject *o = PyLong_FromUnsignedLongLong(v);
    if (!o) throw py::error_already_set();
    return o;
}

py::list tuples3(const Owned &o) {  // bindings.cpp:74-76
    py::list out(o.z);
    for (size_t i = 0; i < o.z; ++i) {
        PyObject *t = PyTuple_New(3);
        if (!t) throw py::error_already_set();
        PyList_SET_ITEM(out.ptr(), (Py_ssize_t)i, t);  // (the list owns it from here, also on an exception)
        PyTuple_SET_ITEM(t, 0, py_u64(o.f[i].start));
        PyTuple_SET_ITEM(t, 1, py_u64(o.f[i].length));
        PyTuple_SET_ITEM(t, 2, py_u64(o.f[i].ref));
    }
    return out;
}

py::list tuples4(const Owned &o) {  // bindings.cpp:226: (start, length, ref without the mask, is_rc)
    py::list out(o.z);
    for (size_t i = 0; i < o.z; ++i) {
        PyObject *t = PyTuple_New(4);
        if (!t) throw py::error_already_set();
        PyList_SET_ITEM(out.ptr(), (Py_ssize_t)i, t);
        PyTuple_SET_ITEM(t, 0, py_u64(o.f[i].start));
        PyTuple_SET_ITEM(t, 1, py_u64(o.f[i].length));
        PyTuple_SET_ITEM(t, 2, py_u64(o.f[i].ref & ~NOLZSS_RC_MASK));
        PyObject *flag = (o.f[i].ref & NOLZSS_RC_MASK) ? Py_True : Py_False;
        Py_INCREF(flag);
        PyTuple_SET_ITEM(t, 3, flag);
    }
    return out;
}

struct PyFactor {  // py::class_<Factor>, bindings.cpp:44-48
    uint64_t start = 0, length = 0, ref = 0;
};

}  // namespace

PYBIND11_MODULE(_noLZSS, m) {
    m.doc() = "Non-overlapping Lempel-Ziv-Storer-Szymanski factorization on MI355X (gfx950): the compiled "
              "module of the reference package, bound to libnolzss_hip.so";

    // (diagnostics) n synthetic factors -> tuples: the cost of the Python-visible result alone
    m.def("_debug_tuples", [](size_t n, int kind) {
        Owned o;
        o.f = static_cast<nolzss_factor *>(std::malloc(sizeof(nolzss_factor) * (n ? n : 1)));
        if (!o.f) throw std::bad_alloc();
        o.z = n;
        for (size_t i = 0; i < n; ++i) o.f[i] = nolzss_factor{i * 20, 1 + i % 40, (i * 2654435761ull) % (i * 20 + 1)};
        if (kind == 2) {
            py::list out(o.z);
            for (size_t i = 0; i < o.z; ++i)
                PyList_SET_ITEM(out.ptr(), (Py_ssize_t)i, py::make_tuple(o.f[i].start, o.f[i].length, o.f[i].ref).release().ptr());
            return out;
        }
        return kind == 1 ? tuples4(o) : tuples3(o);
    }, py::arg("n"), py::arg("kind") = 0);

    py::class_<PyFactor>(m, "Factor", "A factor: start, length, ref (RC_MASK stripped), is_rc")
        .def(py::init<>())
        .def_readonly("start", &PyFactor::start)
        .def_readonly("length", &PyFactor::length)
        .def_property_readonly("ref", [](const PyFactor &f) { return f.ref & ~NOLZSS_RC_MASK; })
        .def_property_readonly("is_rc", [](const PyFactor &f) { return (f.ref & NOLZSS_RC_MASK) != 0; });

    // extension: which HIP device the calls below use (default NOLZSS_DEVICE / LOCAL_RANK / 0)
    if (const char *e = std::getenv("NOLZSS_DEVICE")) g_device = std::atoi(e);
    else if (const char *e2 = std::getenv("LOCAL_RANK")) g_device = std::atoi(e2);
    m.def("set_device", [](int d) { g_device = d; }, py::arg("device"));
    m.def("get_device", [] { return g_device; });
    m.def("device_count", [] {
        int c = 0;
        nolzss_device_count(&c);
        return c;
    });

    m.def("factorize", [](py::buffer b) {  // bindings.cpp:56-77
        const Bytes t = view(b, "factorize");
        Owned o;
        int rc;
        {
            py::gil_scoped_release release;
            rc = nolzss_factorize(t.ptr, t.n, 0, g_device, &o.f, &o.z);
        }
        if (rc) raise_last(rc);
        return tuples3(o);
    }, py::arg("data"), "Factorize a bytes-like object into (start, length, ref) tuples.");

    m.def("factorize_file", [](const std::string &path, size_t /*reserve_hint*/) {  // bindings.cpp:96-105
        Owned o;
        int rc;
        {
            py::gil_scoped_release release;
            rc = nolzss_factorize_file(path.c_str(), 0, g_device, &o.f, &o.z);
        }
        if (rc) raise_last(rc);
        return tuples3(o);
    }, py::arg("path"), py::arg("reserve_hint") = 0);

    m.def("count_factors", [](py::buffer b) {  // bindings.cpp:122-141
        const Bytes t = view(b, "count_factors");
        size_t z = 0;
        int rc;
        {
            py::gil_scoped_release release;
            rc = nolzss_count_factors(t.ptr, t.n, 0, g_device, &z);
        }
        if (rc) raise_last(rc);
        return z;
    }, py::arg("data"));

    m.def("count_factors_file", [](const std::string &path) {  // bindings.cpp:157-164
        size_t z = 0;
        int rc;
        {
            py::gil_scoped_release release;
            rc = nolzss_count_factors_file(path.c_str(), 0, g_device, &z);
        }
        if (rc) raise_last(rc);
        return z;
    }, py::arg("path"));

    m.def("write_factors_binary_file", [](const std::string &in_path, const std::string &out_path) {  // :180-187
        size_t z = 0;
        int rc;
        {
            py::gil_scoped_release release;
            rc = nolzss_write_factors_binary_file(in_path.c_str(), out_path.c_str(), g_device, &z);
        }
        if (rc) raise_last(rc);
        return z;
    }, py::arg("in_path"), py::arg("out_path"));

    m.def("factorize_dna_w_rc", [](py::buffer b) {  // bindings.cpp:207-228
        const Bytes t = view(b, "factorize_dna_w_rc");
        Owned o;
        int rc;
        {
            py::gil_scoped_release release;
            rc = nolzss_factorize_dna_w_rc(t.ptr, t.n, g_device, &o.f, &o.z);
        }
        if (rc) raise_last(rc);
        return tuples4(o);
    }, py::arg("data"));

    m.def("count_factors_dna_w_rc", [](py::buffer b) {  // bindings.cpp:276-295
        const Bytes t = view(b, "count_factors_dna_w_rc");
        size_t z = 0;
        int rc;
        {
            py::gil_scoped_release release;
            rc = nolzss_count_factors_dna_w_rc(t.ptr, t.n, g_device, &z);
        }
        if (rc) raise_last(rc);
        return z;
    }, py::arg("data"));

    m.def("factorize_multiple_dna_w_rc", [](py::buffer b) {  // bindings.cpp:361-382
        const Bytes t = view(b, "factorize_multiple_dna_w_rc");
        Owned o;
        int rc;
        {
            py::gil_scoped_release release;
            rc = nolzss_factorize_multiple_dna_w_rc(t.ptr, t.n, 0, g_device, &o.f, &o.z);
        }
        if (rc) raise_last(rc);
        return tuples4(o);
    }, py::arg("data"));

    m.def("count_factors_multiple_dna_w_rc", [](py::buffer b) {  // bindings.cpp:427-446
        const Bytes t = view(b, "count_factors_multiple_dna_w_rc");
        size_t z = 0;
        int rc;
        {
            py::gil_scoped_release release;
            rc = nolzss_count_factors_multiple_dna_w_rc(t.ptr, t.n, 0, g_device, &z);
        }
        if (rc) raise_last(rc);
        return z;
    }, py::arg("data"));

    m.def("prepare_multiple_dna_sequences_w_rc", [](const std::vector<std::string> &sequences) {  // :732-740
        std::vector<const char *> p;
        std::vector<size_t> l;
        for (const auto &s : sequences) {
            p.push_back(s.data());
            l.push_back(s.size());
        }
        uint8_t *S = nullptr;
        uint64_t *sp = nullptr;
        size_t S_len = 0, orig = 0, ns = 0;
        int rc;
        {
            py::gil_scoped_release release;
            rc = nolzss_prepare_multiple_dna_w_rc(p.data(), l.data(), sequences.size(), &S, &S_len, &orig, &sp, &ns);
        }
        if (rc) raise_last(rc);
        std::string prepared(reinterpret_cast<char *>(S), S_len);
        std::vector<size_t> sentinels(sp, sp + ns);
        nolzss_free(S);
        nolzss_free(sp);
        // (a std::string goes to Python as str, decoded as UTF-8, exactly as in the reference: sentinel
        // bytes >= 128 make this raise UnicodeDecodeError there too)
        return py::make_tuple(prepared, orig, sentinels);
    }, py::arg("sequences"));

    m.def("factorize_w_reference", [](const std::string &reference_seq, const std::string &target_seq) {  // :868-880
        Owned o;
        int rc;
        {
            py::gil_scoped_release release;
            rc = nolzss_factorize_w_reference(reinterpret_cast<const uint8_t *>(reference_seq.data()), reference_seq.size(),
                                              reinterpret_cast<const uint8_t *>(target_seq.data()), target_seq.size(),
                                              g_device, &o.f, &o.z);
        }
        if (rc) raise_last(rc);
        return tuples3(o);
    }, py::arg("reference_seq"), py::arg("target_seq"));

    m.def("factorize_w_reference_file", [](const std::string &reference_seq, const std::string &target_seq,
                                           const std::string &out_path) {
        size_t z = 0;
        int rc;
        {
            py::gil_scoped_release release;
            rc = nolzss_factorize_w_reference_file(reinterpret_cast<const uint8_t *>(reference_seq.data()), reference_seq.size(),
                                                   reinterpret_cast<const uint8_t *>(target_seq.data()), target_seq.size(),
                                                   out_path.c_str(), g_device, &z);
        }
        if (rc) raise_last(rc);
        return z;
    }, py::arg("reference_seq"), py::arg("target_seq"), py::arg("out_path"));

    m.def("factorize_dna_w_reference_seq", [](const std::string &reference_seq, const std::string &target_seq) {  // :800-808
        Owned o;
        int rc;
        {
            py::gil_scoped_release release;
            rc = nolzss_factorize_dna_w_reference_seq(reference_seq.data(), reference_seq.size(), target_seq.data(),
                                                      target_seq.size(), g_device, &o.f, &o.z);
        }
        if (rc) raise_last(rc);
        return tuples4(o);
    }, py::arg("reference_seq"), py::arg("target_seq"));

    m.attr("__version__") = nolzss_version();  // bindings.cpp:1513-1517
}
