/* _rq_fast - the one thing the ctypes veneer (raptor_amd/_lib.py) cannot do fast by itself: the address of a NumPy array's data.
 * `a.ctypes.data` builds a helper object per call (0.9 us), the buffer protocol through ctypes takes 0.4 us and refuses strided or
 * read-only arrays (the README loop hands `observation[:, :22]`, a strided view, to evaluate_step: README.md:97); here it is one
 * PyObject_GetBuffer - 0.06 us for any array.  With four arrays crossing the boundary per iteration that was a quarter of the README
 * loop at the reference's own batch.  Optional: without this module the veneer falls back to the slower ways.  Nothing of the rollout
 * path lives here - libraptor_quad.so does not know Python. */
#define PY_SSIZE_T_CLEAN
#include <Python.h>

static PyObject* rq_address(PyObject* self, PyObject* obj) {
    Py_buffer view;
    (void)self;
    if (PyObject_GetBuffer(obj, &view, PyBUF_STRIDED_RO) != 0) return NULL;
    PyObject* out = PyLong_FromVoidPtr(view.buf);
    PyBuffer_Release(&view);
    return out;
}

static PyMethodDef methods[] = {
    {"address", rq_address, METH_O, "address(array) -> int: where the array's first element lives (any buffer, strided or read-only)"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef module = {PyModuleDef_HEAD_INIT, "_rq_fast", "fast helpers of the raptor_amd ctypes veneer", -1, methods,
                                    NULL, NULL, NULL, NULL};

PyMODINIT_FUNC PyInit__rq_fast(void) { return PyModule_Create(&module); }
