/* _hostfast: the host tail of a predict() batch as one C call (CPython extension, no device code).
 *
 * predict_batch returns, per query, a list of (label, score) tuples (reference classifier.py:1376-1384).  The device stage
 * leaves them packed -- n[b] int32 | class[b, kk] int32 | score[b, kk] float64 (classifier.py::_blend_device) -- and turning
 * 256 x 4 packed hits into 1024 tuples in 256 lists costs ~65 us of interpreter time with the GPU idle behind it (numpy
 * fancy-indexing, two tolist(), zip, 256 slices).  Here: one pass, ~12 us.  classifier.py::_unpack keeps the Python form for an
 * installation without this module (host-side convenience only; the compute path has no fallback).
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>
#include <string.h>
#include <math.h>

/* unpack(names: tuple, packed: buffer, b, kk, off_cls, off_val, kcap) -> (list[list[(label, float)]], has_nan) */
static PyObject* hf_unpack(PyObject* self, PyObject* args) {
    PyObject* names;
    Py_buffer buf;
    Py_ssize_t b, kk, off_cls, off_val, kcap;
    if (!PyArg_ParseTuple(args, "O!y*nnnnn", &PyTuple_Type, &names, &buf, &b, &kk, &off_cls, &off_val, &kcap)) return NULL;
    const Py_ssize_t C = PyTuple_GET_SIZE(names);
    PyObject* out = NULL;
    if (b < 0 || kk < 1 || kcap < 0 || C < 1 || off_cls < 4 * b || off_val < off_cls + 4 * b * kk || (off_val & 7) ||
        buf.len < off_val + 8 * b * kk) {
        PyErr_SetString(PyExc_ValueError, "_hostfast.unpack: layout does not fit the buffer");
        goto done;
    }
    {
        const char* base = (const char*)buf.buf;
        const int32_t* n = (const int32_t*)base;
        const int32_t* cls = (const int32_t*)(base + off_cls);
        const double* val = (const double*)(base + off_val);
        int has_nan = 0;
        for (Py_ssize_t i = 0; i < b * kk; ++i) has_nan |= isnan(val[i]);
        out = PyList_New(b);
        if (!out) goto done;
        for (Py_ssize_t q = 0; q < b; ++q) {
            Py_ssize_t m = n[q] < 0 ? 0 : n[q];
            if (m > kk) m = kk;
            if (m > kcap) m = kcap;
            PyObject* row = PyList_New(m);
            if (!row) { Py_CLEAR(out); goto done; }
            PyList_SET_ITEM(out, q, row);
            for (Py_ssize_t j = 0; j < m; ++j) {
                int32_t c = cls[q * kk + j];
                if (c < 0) c = 0;
                if (c > C - 1) c = (int32_t)(C - 1);
                PyObject* f = PyFloat_FromDouble(val[q * kk + j]);
                PyObject* t = f ? PyTuple_New(2) : NULL;
                if (!t) { Py_XDECREF(f); Py_CLEAR(out); goto done; }
                PyObject* name = PyTuple_GET_ITEM(names, c);
                Py_INCREF(name);
                PyTuple_SET_ITEM(t, 0, name);
                PyTuple_SET_ITEM(t, 1, f);
                PyList_SET_ITEM(row, j, t);
            }
        }
        PyObject* res = Py_BuildValue("(NO)", out, has_nan ? Py_True : Py_False);
        out = res;
    }
done:
    PyBuffer_Release(&buf);
    return out;
}

static PyMethodDef hf_methods[] = {
    {"unpack", hf_unpack, METH_VARARGS, "packed device result -> list of (label, score) lists, has_nan"},
    {NULL, NULL, 0, NULL}};
static struct PyModuleDef hf_module = {PyModuleDef_HEAD_INIT, "_hostfast", "host tail of predict() in C", -1, hf_methods};
PyMODINIT_FUNC PyInit__hostfast(void) { return PyModule_Create(&hf_module); }
