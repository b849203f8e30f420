/* tools/h5_roundtrip.c -- test client of the HDF5 filter plugin (sqeazy_amd/lib/libh5sqy_amd.so), plain HDF5 C API:
 *   h5_roundtrip <raw file> <z> <y> <x> <uint8|uint16> <pipeline> <out.h5> <chunk dump>
 * writes the stack as one chunk (what the reference does, src/hdf5_utils.hpp:723-738) through filter 711 with the header in
 * cd_values, dumps the stored chunk bytes, reads the dataset back through the filter and compares.  The plugin is found
 * through HDF5_PLUGIN_PATH (set by the caller).  Exit code 0 = round trip equal. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "hdf5.h"
#include "../include/sqeazy_amd.h"

int main(int argc, char** argv)
{
    if (argc < 9) { fprintf(stderr, "usage: h5_roundtrip raw z y x dtype pipeline out.h5 chunk.bin\n"); return 2; }
    const hsize_t dims[3] = {(hsize_t)atol(argv[2]), (hsize_t)atol(argv[3]), (hsize_t)atol(argv[4])};
    const int voxel = strcmp(argv[5], "uint8") == 0 ? 1 : 2;
    const size_t nbytes = (size_t)dims[0] * dims[1] * dims[2] * (size_t)voxel;
    char* raw = (char*)malloc(nbytes), *back = (char*)calloc(nbytes, 1);
    FILE* f = fopen(argv[1], "rb");
    if (!f || fread(raw, 1, nbytes, f) != nbytes) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
    fclose(f);

    long shape[3] = {(long)dims[0], (long)dims[1], (long)dims[2]}, hlen = 0;
    if (SQYAMD_Header_Build(argv[6], voxel, shape, 3, 0, NULL, &hlen)) { fprintf(stderr, "cannot build header\n"); return 2; }
    const size_t ncd = ((size_t)hlen + sizeof(unsigned) - 1) / sizeof(unsigned);
    unsigned* cd = (unsigned*)calloc(ncd, sizeof(unsigned));
    long cap = (long)(ncd * sizeof(unsigned));
    if (SQYAMD_Header_Build(argv[6], voxel, shape, 3, 0, (char*)cd, &cap)) return 2;

    if (H5Zfilter_avail(01307) <= 0) { fprintf(stderr, "filter 711 not available (HDF5_PLUGIN_PATH?)\n"); return 3; }
    const hid_t type = voxel == 1 ? H5T_NATIVE_UINT8 : H5T_NATIVE_UINT16;
    hid_t file = H5Fcreate(argv[7], H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
    hid_t space = H5Screate_simple(3, dims, NULL);
    hid_t dcpl = H5Pcreate(H5P_DATASET_CREATE);
    H5Pset_chunk(dcpl, 3, dims);
    if (H5Pset_filter(dcpl, 01307, H5Z_FLAG_MANDATORY, ncd, cd) < 0) return 3;
    hid_t ds = H5Dcreate2(file, "sqy_stack", type, space, H5P_DEFAULT, dcpl, H5P_DEFAULT);
    /* the filter runs when the chunk leaves the cache, i.e. possibly only at close */
    if (ds < 0 || H5Dwrite(ds, type, H5S_ALL, H5S_ALL, H5P_DEFAULT, raw) < 0 || H5Dclose(ds) < 0 || H5Fclose(file) < 0) {
        fprintf(stderr, "write through the filter failed\n");
        return 4;
    }

    file = H5Fopen(argv[7], H5F_ACC_RDONLY, H5P_DEFAULT);
    ds = H5Dopen2(file, "sqy_stack", H5P_DEFAULT);
    const hsize_t origin[3] = {0, 0, 0};
    hsize_t stored = 0;
    H5Dget_chunk_storage_size(ds, origin, &stored);
    char* chunk = (char*)malloc((size_t)stored);
    uint32_t mask = 0;
    if (H5Dread_chunk(ds, H5P_DEFAULT, origin, &mask, chunk) < 0) return 5;
    f = fopen(argv[8], "wb"); fwrite(chunk, 1, (size_t)stored, f); fclose(f);
    if (H5Dread(ds, type, H5S_ALL, H5S_ALL, H5P_DEFAULT, back) < 0) { fprintf(stderr, "read through the filter failed\n"); return 6; }
    H5Dclose(ds); H5Fclose(file);
    printf("raw %zu bytes, stored chunk %llu bytes, round trip %s\n", nbytes, (unsigned long long)stored, memcmp(raw, back, nbytes) == 0 ? "equal" : "DIFFERENT");
    return memcmp(raw, back, nbytes) == 0 ? 0 : 1;
}
