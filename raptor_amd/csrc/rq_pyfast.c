/* _rq_fast - what the ctypes veneer (raptor_amd/_lib.py, l2f.py, foundation_policy.py) cannot do fast by itself.
 *
 *  address(array)      the address of a NumPy array's data.  `a.ctypes.data` builds a helper object per call (0.9 us), the buffer
 *                      protocol through ctypes takes 0.4 us and refuses strided or read-only arrays (the README loop hands
 *                      `observation[:, :22]`, a strided view, to evaluate_step: README.md:97); here it is one PyObject_GetBuffer -
 *                      0.06 us for any array.
 *  observe / evaluate_step / step / assign / rollout
 *                      the four calls of the reference's loop (README.md:96-99) without the ctypes foreign-call machinery: an
 *                      eight-argument call through ctypes costs 0.7 us before the library sees it, and at the reference's own batch
 *                      (8 envs) the library's own work per call is of that order (DESIGN.md section 5 "Resident executor").  Each takes the
 *                      ADDRESS of the C entry point (an int the veneer got from ctypes once), the handles as the veneer holds them
 *                      (ctypes c_void_p objects, ints or None) and the arrays themselves, checks dtype / shape / layout on the buffer
 *                      (what the Python code checked attribute by attribute), releases the GIL around the call like ctypes does, and
 *                      returns the library's status (0 = ok), or NOT_HANDLED (1) without having called anything when an argument is
 *                      not what the fast path takes - the veneer then goes the ordinary way, which also words the error.
 *
 * Optional: without this module the veneer falls back to ctypes for everything.  Nothing of the rollout path lives here -
 * libraptor_quad.so does not know Python, and this file does not link against it (signatures: include/raptor_quad.h rq_observe,
 * rq_policy_evaluate_step, rq_step, rq_state_assign, rq_rollout; tests/test_capi_cpu.py holds them to the header). */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>

#define NOT_HANDLED 1

typedef int (*observe_fn)(void*, void*, const void*, const void*, float*, void*);
typedef int (*evaluate_step_fn)(void*, void*, const float*, uint32_t, uint32_t, float*);
typedef int (*step_fn)(void*, void*, const void*, const void*, const float*, void*, void*, float*);
typedef int (*assign_fn)(void*, const void*);
typedef int (*rollout_fn)(void*, void*, const void*, void*, void*, void*, uint32_t, int, uint32_t);

static PyObject* rq_address(PyObject* self, PyObject* obj) {
    Py_buffer view;
    (void)self;
    if (PyObject_GetBuffer(obj, &view, PyBUF_STRIDED_RO) != 0) return NULL;
    PyObject* out = PyLong_FromVoidPtr(view.buf);
    PyBuffer_Release(&view);
    return out;
}

/* a handle as the veneer holds it: a ctypes c_void_p (its buffer IS the pointer), an int, or None (-> not handled).  1 = ok */
static int handle_of(PyObject* obj, void** out) {
    if (obj == Py_None) return 0;
    if (PyLong_Check(obj)) {
        *out = PyLong_AsVoidPtr(obj);
        if (*out == NULL) { PyErr_Clear(); return 0; }
        return 1;
    }
    Py_buffer view;
    if (PyObject_GetBuffer(obj, &view, PyBUF_SIMPLE) != 0) { PyErr_Clear(); return 0; }
    const int ok = view.len == (Py_ssize_t)sizeof(void*) && *(void**)view.buf != NULL;
    if (ok) *out = *(void**)view.buf;
    PyBuffer_Release(&view);
    return ok;
}

static int is_f32(const Py_buffer* v) {
    return v->itemsize == 4 && v->format != NULL && ((v->format[0] == 'f' && v->format[1] == 0) ||
                                                      ((v->format[0] == '<' || v->format[0] == '=') && v->format[1] == 'f' && v->format[2] == 0));
}

/* a float32 array [rows, cols] whose rows are contiguous; *stride = floats between rows (>= cols).  1 = ok (view held) */
static int rows_of(PyObject* obj, int writable, Py_ssize_t rows, Py_ssize_t min_cols, Py_ssize_t exact_cols, int need_contiguous,
                   Py_buffer* view, Py_ssize_t* stride) {
    if (PyObject_GetBuffer(obj, view, (writable ? PyBUF_WRITABLE : 0) | PyBUF_STRIDES | PyBUF_FORMAT) != 0) { PyErr_Clear(); return 0; }
    int ok = view->ndim == 2 && is_f32(view) && (rows < 0 || view->shape[0] == rows) && view->shape[1] >= min_cols &&
             (exact_cols < 0 || view->shape[1] == exact_cols) && view->strides[1] == 4 && view->strides[0] % 4 == 0 &&
             view->strides[0] >= 4 * view->shape[1] && view->shape[0] > 0;
    if (ok && need_contiguous) ok = view->strides[0] == 4 * view->shape[1];
    if (!ok) { PyBuffer_Release(view); return 0; }
    *stride = view->strides[0] / 4;
    return 1;
}

static PyObject* not_handled(void) { return PyLong_FromLong(NOT_HANDLED); }

/* observe(fn, device, env, params, state, observation [n, dim] float32 C-contiguous writable, rng, n, dim) -> status */
static PyObject* rq_fast_observe(PyObject* self, PyObject* const* a, Py_ssize_t nargs) {
    (void)self;
    if (nargs != 9) { PyErr_SetString(PyExc_TypeError, "observe takes 9 arguments"); return NULL; }
    void *fn, *dev, *env, *params, *state, *rng;
    if (!handle_of(a[0], &fn) || !handle_of(a[1], &dev) || !handle_of(a[2], &env) || !handle_of(a[3], &params) ||
        !handle_of(a[4], &state) || !handle_of(a[6], &rng))
        return not_handled();
    const Py_ssize_t n = PyLong_AsSsize_t(a[7]), dim = PyLong_AsSsize_t(a[8]);
    if (PyErr_Occurred()) return NULL;
    Py_buffer obs; Py_ssize_t stride;
    if (!rows_of(a[5], 1, n, dim, dim, 1, &obs, &stride)) return not_handled();
    int status;
    Py_BEGIN_ALLOW_THREADS
    status = ((observe_fn)fn)(dev, env, params, state, (float*)obs.buf, rng);
    Py_END_ALLOW_THREADS
    PyBuffer_Release(&obs);
    return PyLong_FromLong(status);
}

/* evaluate_step(fn, policy, observation [batch, >= min_cols] float32 rows, action [batch, 4] float32 C-contiguous writable, min_cols)
 * -> status; the observation's rows may be strided (observation[:, :22] of a wider array, README.md:97) */
static PyObject* rq_fast_evaluate_step(PyObject* self, PyObject* const* a, Py_ssize_t nargs) {
    (void)self;
    if (nargs != 5) { PyErr_SetString(PyExc_TypeError, "evaluate_step takes 5 arguments"); return NULL; }
    void *fn, *pol;
    if (!handle_of(a[0], &fn) || !handle_of(a[1], &pol)) return not_handled();
    const Py_ssize_t min_cols = PyLong_AsSsize_t(a[4]);
    if (PyErr_Occurred()) return NULL;
    Py_buffer obs, act; Py_ssize_t so, sa;
    if (!rows_of(a[2], 0, -1, min_cols, -1, 0, &obs, &so)) return not_handled();
    if (so > 0xFFFFFFFFll || obs.shape[0] > 0xFFFFFFFFll || !rows_of(a[3], 1, obs.shape[0], 4, 4, 1, &act, &sa)) {
        PyBuffer_Release(&obs);
        return not_handled();
    }
    int status;
    Py_BEGIN_ALLOW_THREADS
    status = ((evaluate_step_fn)fn)(pol, NULL, (const float*)obs.buf, (uint32_t)obs.shape[0], (uint32_t)so, (float*)act.buf);
    Py_END_ALLOW_THREADS
    PyBuffer_Release(&obs);
    PyBuffer_Release(&act);
    return PyLong_FromLong(status);
}

/* step(fn, device, env, params, state, action [n, 4] float32 C-contiguous, next_state, rng, n) -> status (dts = NULL) */
static PyObject* rq_fast_step(PyObject* self, PyObject* const* a, Py_ssize_t nargs) {
    (void)self;
    if (nargs != 9) { PyErr_SetString(PyExc_TypeError, "step takes 9 arguments"); return NULL; }
    void *fn, *dev, *env, *params, *state, *next, *rng;
    if (!handle_of(a[0], &fn) || !handle_of(a[1], &dev) || !handle_of(a[2], &env) || !handle_of(a[3], &params) ||
        !handle_of(a[4], &state) || !handle_of(a[6], &next) || !handle_of(a[7], &rng))
        return not_handled();
    const Py_ssize_t n = PyLong_AsSsize_t(a[8]);
    if (PyErr_Occurred()) return NULL;
    Py_buffer act; Py_ssize_t stride;
    if (!rows_of(a[5], 0, n, 4, 4, 1, &act, &stride)) return not_handled();
    int status;
    Py_BEGIN_ALLOW_THREADS
    status = ((step_fn)fn)(dev, env, params, state, (const float*)act.buf, next, rng, NULL);
    Py_END_ALLOW_THREADS
    PyBuffer_Release(&act);
    return PyLong_FromLong(status);
}

/* assign(fn, dst, src) -> status */
static PyObject* rq_fast_assign(PyObject* self, PyObject* const* a, Py_ssize_t nargs) {
    (void)self;
    if (nargs != 3) { PyErr_SetString(PyExc_TypeError, "assign takes 3 arguments"); return NULL; }
    void *fn, *dst, *src;
    if (!handle_of(a[0], &fn) || !handle_of(a[1], &dst) || !handle_of(a[2], &src)) return not_handled();
    return PyLong_FromLong(((assign_fn)fn)(dst, src));     /* host-side bookkeeping only: the GIL stays */
}

/* rollout(fn, device, env, params, state, policy, rng, n_steps, mode, flags) -> status: rq_rollout, the call inside bench.py's timed
 * region (one launch per region of 20 steps: the call itself is 5 % of it) */
static PyObject* rq_fast_rollout(PyObject* self, PyObject* const* a, Py_ssize_t nargs) {
    (void)self;
    if (nargs != 10) { PyErr_SetString(PyExc_TypeError, "rollout takes 10 arguments"); return NULL; }
    void *fn, *dev, *env, *params, *state, *policy, *rng;
    if (!handle_of(a[0], &fn) || !handle_of(a[1], &dev) || !handle_of(a[2], &env) || !handle_of(a[3], &params) ||
        !handle_of(a[4], &state) || !handle_of(a[5], &policy) || !handle_of(a[6], &rng))
        return not_handled();
    const unsigned long n_steps = PyLong_AsUnsignedLong(a[7]);
    const long mode = PyLong_AsLong(a[8]);
    const unsigned long flags = PyLong_AsUnsignedLong(a[9]);
    if (PyErr_Occurred()) return NULL;
    if (n_steps > 0xFFFFFFFFul || flags > 0xFFFFFFFFul || mode < -0x7FFFFFFFl || mode > 0x7FFFFFFFl) return not_handled();
    int status;
    Py_BEGIN_ALLOW_THREADS
    status = ((rollout_fn)fn)(dev, env, params, state, policy, rng, (uint32_t)n_steps, (int)mode, (uint32_t)flags);
    Py_END_ALLOW_THREADS
    return PyLong_FromLong(status);
}

static PyMethodDef methods[] = {
    {"address", rq_address, METH_O, "address(array) -> int: where the array's first element lives (any buffer, strided or read-only)"},
    {"observe", (PyCFunction)(void (*)(void))rq_fast_observe, METH_FASTCALL, "rq_observe with a host array; -> status, 1 = not handled"},
    {"evaluate_step", (PyCFunction)(void (*)(void))rq_fast_evaluate_step, METH_FASTCALL,
     "rq_policy_evaluate_step with host arrays; -> status, 1 = not handled"},
    {"step", (PyCFunction)(void (*)(void))rq_fast_step, METH_FASTCALL, "rq_step with a host action array; -> status, 1 = not handled"},
    {"assign", (PyCFunction)(void (*)(void))rq_fast_assign, METH_FASTCALL, "rq_state_assign; -> status, 1 = not handled"},
    {"rollout", (PyCFunction)(void (*)(void))rq_fast_rollout, METH_FASTCALL, "rq_rollout; -> status, 1 = not handled"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef module = {PyModuleDef_HEAD_INIT, "_rq_fast", "fast helpers of the raptor_amd ctypes veneer", -1, methods,
                                    NULL, NULL, NULL, NULL};

PyMODINIT_FUNC PyInit__rq_fast(void) {
    PyObject* m = PyModule_Create(&module);
    if (m != NULL) PyModule_AddIntConstant(m, "NOT_HANDLED", NOT_HANDLED);
    return m;
}
