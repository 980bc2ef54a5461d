/* Writes tests/golden/caffe_vgg_tiny.h5 with the REAL HDF5 library (libhdf5 1.10.6 + hdf5_hl of the build image, default
 * "earliest" file format = what Caffe's HDF5 snapshots use): the layout of the reference's vgg16.caffemodel.h5
 * (/data/<layer>/{0,1}, /data/batch_<layer>/{0,1,2}; imm/models/selfsup/vgg16.py:17-47,74-92) at toy sizes, so that
 * imm_amd/utils/hdf5_lite.py (pure Python) is tested against a file it did not write.  Also a few datasets that exercise
 * the optional paths: chunked + shuffle + deflate, chunked unfiltered with ragged edge chunks, float64, int32, a scalar, a
 * compact dataset, an attribute (ignored by the reader).
 *
 * Every value is f(layer index, blob index, element index) = sin(0.37 * e + 1.3 * b + 0.11 * l), so the test recomputes the
 * expected arrays instead of storing them.
 *
 *   gcc tests/golden/make_h5_golden.c -I/opt/conda/include -L/opt/conda/lib -Wl,-rpath,/opt/conda/lib -lhdf5_hl -lhdf5 -lm \
 *       -o /tmp/make_h5_golden && /tmp/make_h5_golden tests/golden/caffe_vgg_tiny.h5
 *
 * `make_h5_golden <out.h5> full` writes the REAL shapes of the colourisation VGG16 instead (conv1_1 [64,1,3,3] ... conv5_3
 * [512,512,3,3], 59 MB; every layer with its batch_<layer> group): blob values are a counter-based hash (hash_unit below,
 * restated in numpy by tests/test_vgg_file_gpu.py) scaled He-style, so that a network built from the file has healthy
 * activations.  That file is generated at test time (it is too large to commit) and feeds
 * hdf5_lite -> load_vgg16 -> engine on the GPU against the oracle with the same folded weights.
 */
#include <hdf5.h>
#include <hdf5_hl.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

static float val(int l, int b, long e) { return (float)sin(0.37 * (double)e + 1.3 * b + 0.11 * l); }

/* uniform in [0,1): 32-bit mix of (element, layer, blob) */
static double hash_unit(int l, int b, long e) {
  unsigned int h = (unsigned int)e * 2654435761u + (unsigned int)l * 97u + (unsigned int)b * 13u + 12345u;
  h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
  return (double)h / 4294967296.0;
}

static void hash_blob(hid_t g, const char* name, int rank, const hsize_t* dims, int l, int b, double lo, double hi) {
  long n = 1;
  for (int i = 0; i < rank; ++i) n *= (long)dims[i];
  float* buf = (float*)malloc(sizeof(float) * n);
  for (long e = 0; e < n; ++e) buf[e] = (float)(lo + (hi - lo) * hash_unit(l, b, e));
  H5LTmake_dataset_float(g, name, rank, dims, buf);
  free(buf);
}

static int write_full(const char* path) {
  static const char* names[13] = {"conv1_1", "conv1_2", "conv2_1", "conv2_2", "conv3_1", "conv3_2", "conv3_3",
                                  "conv4_1", "conv4_2", "conv4_3", "conv5_1", "conv5_2", "conv5_3"};
  static const int couts[13] = {64, 64, 128, 128, 256, 256, 256, 512, 512, 512, 512, 512, 512};
  hid_t f = H5Fcreate(path, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
  hid_t data = H5Gcreate2(f, "data", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
  int cin = 1;
  for (int l = 0; l < 13; ++l) {
    const int cout = couts[l];
    hid_t g = H5Gcreate2(data, names[l], H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    hsize_t wd[4] = {(hsize_t)cout, (hsize_t)cin, 3, 3}, bd[1] = {(hsize_t)cout}, one[1] = {1};
    const double a = sqrt(6.0 / (9.0 * cin));          /* uniform(-a, a): variance 2 / fan_in (He) */
    hash_blob(g, "0", 4, wd, l, 0, -a, a);
    hash_blob(g, "1", 1, bd, l, 1, -0.1, 0.1);
    H5Gclose(g);
    char bn[64];
    snprintf(bn, sizeof bn, "batch_%s", names[l]);
    g = H5Gcreate2(data, bn, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    const float scale = 2.0f + 0.25f * (float)l;      /* caffe's accumulated scale factor: sums are scale x the statistic */
    hash_blob(g, "0", 1, bd, l, 2, -0.05 * scale, 0.05 * scale);   /* mean sums */
    hash_blob(g, "1", 1, bd, l, 3, 0.5 * scale, 1.5 * scale);      /* variance sums (positive) */
    H5LTmake_dataset_float(g, "2", 1, one, &scale);
    H5Gclose(g);
    cin = cout;
  }
  H5Gclose(data);
  H5Fclose(f);
  return 0;
}

static void blob(hid_t g, const char* name, int rank, const hsize_t* dims, int l, int b) {
  long n = 1;
  for (int i = 0; i < rank; ++i) n *= (long)dims[i];
  float* buf = (float*)malloc(sizeof(float) * n);
  for (long e = 0; e < n; ++e) buf[e] = val(l, b, e);
  H5LTmake_dataset_float(g, name, rank, dims, buf); /* contiguous layout, like Caffe's hdf5_save_nd_dataset */
  free(buf);
}

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  if (argc > 2 && argv[2][0] == 'f') return write_full(argv[1]);
  hid_t f = H5Fcreate(argv[1], H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
  hid_t data = H5Gcreate2(f, "data", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
  /* 13 conv layers with toy channel counts (conv1_1 has ONE input channel: the colourisation VGG is grayscale) + their batch norms:
   * 26 groups under /data -> several symbol-table nodes behind the group's B-tree */
  const char* names[13] = {"conv1_1", "conv1_2", "conv2_1", "conv2_2", "conv3_1", "conv3_2", "conv3_3",
                           "conv4_1", "conv4_2", "conv4_3", "conv5_1", "conv5_2", "conv5_3"};
  int cin = 1;
  for (int l = 0; l < 13; ++l) {
    int cout = 4 + (l % 3);
    hid_t g = H5Gcreate2(data, names[l], H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    hsize_t wd[4] = {(hsize_t)cout, (hsize_t)cin, 3, 3}, bd[1] = {(hsize_t)cout};
    blob(g, "0", 4, wd, l, 0);
    blob(g, "1", 1, bd, l, 1);
    H5Gclose(g);
    char bn[64];
    snprintf(bn, sizeof bn, "batch_%s", names[l]);
    g = H5Gcreate2(data, bn, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    hsize_t one[1] = {1};
    blob(g, "0", 1, bd, l, 2);
    {                                   /* variance sums must be positive */
      float* v = (float*)malloc(sizeof(float) * cout);
      for (int e = 0; e < cout; ++e) v[e] = 2.0f + val(l, 3, e);
      H5LTmake_dataset_float(g, "1", 1, bd, v);
      free(v);
    }
    float scale = 3.0f + (float)l;
    H5LTmake_dataset_float(g, "2", 1, one, &scale);
    H5Gclose(g);
    cin = cout;
  }
  H5LTset_attribute_string(f, "data", "note", "toy caffe snapshot");
  H5Gclose(data);

  hid_t ex = H5Gcreate2(f, "extra", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
  { /* chunked + shuffle + deflate, ragged edge chunks */
    hsize_t dims[2] = {37, 21}, chunk[2] = {8, 8};
    float buf[37 * 21];
    for (int e = 0; e < 37 * 21; ++e) buf[e] = val(20, 0, e);
    hid_t sp = H5Screate_simple(2, dims, NULL), pl = H5Pcreate(H5P_DATASET_CREATE);
    H5Pset_chunk(pl, 2, chunk); H5Pset_shuffle(pl); H5Pset_deflate(pl, 6);
    hid_t d = H5Dcreate2(ex, "chunked_deflate", H5T_IEEE_F32LE, sp, H5P_DEFAULT, pl, H5P_DEFAULT);
    H5Dwrite(d, H5T_NATIVE_FLOAT, H5S_ALL, H5S_ALL, H5P_DEFAULT, buf);
    H5Dclose(d); H5Pclose(pl); H5Sclose(sp);
  }
  { /* chunked, no filter, float64, 3-d */
    hsize_t dims[3] = {5, 6, 7}, chunk[3] = {2, 4, 3};
    double buf[5 * 6 * 7];
    for (int e = 0; e < 5 * 6 * 7; ++e) buf[e] = (double)val(21, 0, e);
    hid_t sp = H5Screate_simple(3, dims, NULL), pl = H5Pcreate(H5P_DATASET_CREATE);
    H5Pset_chunk(pl, 3, chunk);
    hid_t d = H5Dcreate2(ex, "chunked_f64", H5T_IEEE_F64LE, sp, H5P_DEFAULT, pl, H5P_DEFAULT);
    H5Dwrite(d, H5T_NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, buf);
    H5Dclose(d); H5Pclose(pl); H5Sclose(sp);
  }
  { /* int32 contiguous, scalar float, compact dataset, big-endian float */
    hsize_t dims[1] = {9};
    int ib[9];
    for (int e = 0; e < 9; ++e) ib[e] = e * e - 7;
    H5LTmake_dataset_int(ex, "ints", 1, dims, ib);
    float s = 2.5f;
    hid_t sp = H5Screate(H5S_SCALAR);
    hid_t d = H5Dcreate2(ex, "scalar", H5T_IEEE_F32LE, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    H5Dwrite(d, H5T_NATIVE_FLOAT, H5S_ALL, H5S_ALL, H5P_DEFAULT, &s);
    H5Dclose(d); H5Sclose(sp);
    hsize_t cd[1] = {6};
    float cb[6];
    for (int e = 0; e < 6; ++e) cb[e] = val(22, 0, e);
    sp = H5Screate_simple(1, cd, NULL);
    hid_t pl = H5Pcreate(H5P_DATASET_CREATE);
    H5Pset_layout(pl, H5D_COMPACT);
    d = H5Dcreate2(ex, "compact", H5T_IEEE_F32LE, sp, H5P_DEFAULT, pl, H5P_DEFAULT);
    H5Dwrite(d, H5T_NATIVE_FLOAT, H5S_ALL, H5S_ALL, H5P_DEFAULT, cb);
    H5Dclose(d); H5Pclose(pl);
    d = H5Dcreate2(ex, "big_endian", H5T_IEEE_F32BE, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    H5Dwrite(d, H5T_NATIVE_FLOAT, H5S_ALL, H5S_ALL, H5P_DEFAULT, cb);
    H5Dclose(d); H5Sclose(sp);
  }
  H5Gclose(ex);
  H5Fclose(f);
  return 0;
}
