/* fastrank.c -- CPython glue for the reference's online call, ColbertRanker.rank_forward(Q, pids, depth)
 * (colbert/ranking/colbert_ranker.py:75-137 as faiss_indexers.py:234 calls it: a python list of ~1000 pids in, two python
 * lists out).  Host side only, above the C ABI: it turns the list into the pinned input buffer, calls
 * maxsim_rank_forward (include/maxsim.h) through the function pointer colbert_amd/_lib.py resolved, and builds the
 * result lists -- what colbert_amd/ranker.py otherwise does with array('q') + ctypes + numpy, at ~10 us less per call
 * (of ~60).  No compute here, and no second code path for the scores: the library call is the same.
 * Built by colbert_amd/csrc/build.sh with gcc against the interpreter's headers -> colbert_amd/_fastrank.so. */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>

typedef int (*rank_forward_fn)(const void* iv, const void* Q, int q_dtype, int Lq, const int64_t* pids, int n, int depth,
                               void* workspace, int64_t* out_pids, float* out_scores, uint32_t* done_flag, int sync,
                               void* stream);

/* rank_forward(fn, iv, Q, q_dtype, Lq, pids: list[int], depth, pin_in, scratch, out_pids, out_scores, flag, stream, n_docs)
 *   -> (pids: list[int], scores: list[float])  or  int (a negative MAXSIM_E* code: the caller raises)  or  None (the list
 *      holds negative pids in [-n_docs, -1]: torch indexing wraps them, colbert_ranker.py:88 -- the caller's general path does)
 * A pid >= n_docs or < -n_docs raises IndexError, as `self.doclens[pids]` does at colbert_ranker.py:88.
 * All pointers are integers (addresses).  pin_in must hold len(pids) int64. */
static PyObject* fr_rank_forward(PyObject* self, PyObject* args) {
  unsigned long long fn, iv, q, pin_in, scratch, out_p, out_s, flag, stream;
  long long n_docs;
  int q_dtype, Lq, depth;
  PyObject* pids;
  (void)self;
  if (!PyArg_ParseTuple(args, "KKKiiOiKKKKKKL", &fn, &iv, &q, &q_dtype, &Lq, &pids, &depth, &pin_in, &scratch, &out_p, &out_s,
                        &flag, &stream, &n_docs))
    return NULL;
  if (!PyList_CheckExact(pids)) {
    PyErr_SetString(PyExc_TypeError, "pids must be a list");
    return NULL;
  }
  const Py_ssize_t n = PyList_GET_SIZE(pids);
  if (n < 1 || n > 16384 || depth < 1 || !fn || !pin_in || !out_p || !out_s) {
    PyErr_SetString(PyExc_ValueError, "rank_forward: 1 <= len(pids) <= 16384, depth >= 1, non-null buffers");
    return NULL;
  }
  int64_t* const dst = (int64_t*)(uintptr_t)pin_in;
  int64_t lo = INT64_MAX, hi = INT64_MIN;
  for (Py_ssize_t i = 0; i < n; ++i) {
    PyObject* const o = PyList_GET_ITEM(pids, i);
    if (!PyLong_Check(o)) {
      PyErr_SetString(PyExc_TypeError, "pids must be ints");
      return NULL;
    }
#if PY_VERSION_HEX < 0x030C0000 && PYLONG_BITS_IN_DIGIT == 30
    /* non-negative ints below 2^60 straight from their one or two 30-bit digits (every pid of a real index) */
    const Py_ssize_t sz = Py_SIZE(o);
    if ((size_t)sz <= 2) {
      const digit* const dg = ((PyLongObject*)o)->ob_digit;
      const int64_t u = sz == 0 ? 0 : sz == 1 ? (int64_t)dg[0] : (int64_t)dg[0] | ((int64_t)dg[1] << 30);
      dst[i] = u;
      hi = u > hi ? u : hi;
      lo = u < lo ? u : lo;
      continue;
    }
#endif
    const long long v = PyLong_AsLongLong(o);
    if (v == -1 && PyErr_Occurred()) return NULL;
    dst[i] = (int64_t)v;
    hi = v > hi ? v : hi;
    lo = v < lo ? v : lo;
  }
  if (hi >= (int64_t)n_docs || lo < -(int64_t)n_docs) {  /* self.doclens[pids], colbert_ranker.py:88 */
    PyErr_Format(PyExc_IndexError, "index %lld is out of bounds for dimension 0 with size %lld",
                 (long long)(hi >= (int64_t)n_docs ? hi : lo), n_docs);
    return NULL;
  }
  if (lo < 0) Py_RETURN_NONE;
  const int k = depth < (int)n ? depth : (int)n;
  int rc;
  Py_BEGIN_ALLOW_THREADS
  rc = ((rank_forward_fn)(uintptr_t)fn)((const void*)(uintptr_t)iv, (const void*)(uintptr_t)q, q_dtype, Lq, dst, (int)n, depth,
                                        (void*)(uintptr_t)scratch, (int64_t*)(uintptr_t)out_p, (float*)(uintptr_t)out_s,
                                        (uint32_t*)(uintptr_t)flag, 1, (void*)(uintptr_t)stream);
  Py_END_ALLOW_THREADS
  if (rc != 0) return PyLong_FromLong(rc);
  const int64_t* const rp = (const int64_t*)(uintptr_t)out_p;
  const float* const rs = (const float*)(uintptr_t)out_s;
  PyObject* const lp = PyList_New(k);
  PyObject* const ls = PyList_New(k);
  if (!lp || !ls) {
    Py_XDECREF(lp);
    Py_XDECREF(ls);
    return NULL;
  }
  for (int i = 0; i < k; ++i) {
    PyObject* const a = PyLong_FromLongLong((long long)rp[i]);
    PyObject* const b = PyFloat_FromDouble((double)rs[i]);
    if (!a || !b) {
      Py_XDECREF(a);
      Py_XDECREF(b);
      Py_DECREF(lp);
      Py_DECREF(ls);
      return NULL;
    }
    PyList_SET_ITEM(lp, i, a);
    PyList_SET_ITEM(ls, i, b);
  }
  return Py_BuildValue("(NN)", lp, ls);
}

static PyMethodDef fr_methods[] = {
    {"rank_forward", fr_rank_forward, METH_VARARGS,
     "rank_forward(fn, iv, Q, q_dtype, Lq, pids, depth, pin_in, scratch, out_pids, out_scores, flag, stream) -> (pids, scores) | rc"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef fr_module = {PyModuleDef_HEAD_INIT, "_fastrank",
                                       "CPython glue of ColbertRanker.rank_forward's online path (see fastrank.c)", -1,
                                       fr_methods, NULL, NULL, NULL, NULL};

PyMODINIT_FUNC PyInit__fastrank(void) { return PyModule_Create(&fr_module); }
