/* qk_test_source.c -- a minimal C sample-source plug-in, written the way import_quisk_api.c:10-31 tells plug-in authors to:
 * it imports the capsule `_quisk.QUISK_C_API`, takes the function pointers by the indices of quisk.h:326-335, registers
 * start / stop / read callbacks with quisk_sample_source (index 7) and counts into the sound-state record (index 0).
 * Test infrastructure: tests/test_quisk_ext.py loads complex samples into it and lets `_quisk.read_sound()` pull them. */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <complex.h>
#include <string.h>

typedef void (*ty_start)(void);
typedef void (*ty_stop)(void);
typedef int (*ty_read)(double _Complex *);
static void **api;
static double _Complex *data;
static long n_data, pos, chunk = 4800;
static int started, stopped;

static void src_start(void) { started++; }
static void src_stop(void) { stopped++; }
static int src_read(double _Complex *buf)
{   /* hand over the next chunk (the buffer holds SAMP_BUFFER_SIZE = 66000 samples, quisk.h:15) */
    long n = n_data - pos;
    if (n > chunk) n = chunk;
    if (n <= 0) return 0;
    memcpy(buf, data + pos, (size_t)n * sizeof *buf);
    pos += n;
    return (int)n;
}

static PyObject *load(PyObject *self, PyObject *args)
{
    Py_buffer view;
    long c = 4800;
    if (!PyArg_ParseTuple(args, "y*|l", &view, &c)) return NULL;
    free(data);
    n_data = view.len / 16; pos = 0; chunk = c;
    data = malloc((size_t)n_data * 16 + 16);
    memcpy(data, view.buf, (size_t)n_data * 16);
    PyBuffer_Release(&view);
    Py_RETURN_NONE;
}
static PyObject *attach(PyObject *self, PyObject *args)
{   /* quisk_sample_source(start, stop, read): capsule entry 7 */
    ((void (*)(ty_start, ty_stop, ty_read))api[7])(src_start, src_stop, src_read);
    Py_RETURN_NONE;
}
static PyObject *state(PyObject *self, PyObject *args)
{   /* (started, stopped, sample_rate read through entry 0, QuiskGetConfigInt("graph_refresh", -1), QuiskTimeSec() > 0) */
    const int *sound = (const int *)((const char *)api[0] + 256);      /* struct sound_conf: two 128-byte names, then sample_rate */
    const int refresh = ((int (*)(const char *, int))api[1])("graph_refresh", -1);
    const double now = ((double (*)(void))api[4])();
    return Py_BuildValue("iiiii", started, stopped, sound[0], refresh, now > 0);
}
static PyObject *count_read_error(PyObject *self, PyObject *args)
{   /* pt_quisk_sound_state->read_error++ (import_quisk_api.c:27-31): offset of read_error = 256 + 8 * 4 */
    int *read_error = (int *)((char *)api[0] + 256 + 32);
    (*read_error)++;
    Py_RETURN_NONE;
}
static PyMethodDef methods[] = {
    { "load", load, METH_VARARGS, "load(complex128 bytes, chunk)" }, { "attach", attach, METH_NOARGS, "register the callbacks" },
    { "state", state, METH_NOARGS, "" }, { "count_read_error", count_read_error, METH_NOARGS, "" }, { NULL, NULL, 0, NULL } };
static struct PyModuleDef mod = { PyModuleDef_HEAD_INIT, "qk_test_source", NULL, -1, methods };
PyMODINIT_FUNC PyInit_qk_test_source(void)
{
    PyObject *m = PyModule_Create(&mod);
    if (!m) return NULL;
    api = (void **)PyCapsule_Import("_quisk.QUISK_C_API", 0);
    if (!api) { Py_DECREF(m); return NULL; }
    return m;
}
