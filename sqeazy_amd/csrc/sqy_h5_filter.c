/* sqy_h5_filter.c -- HDF5 dynamically loaded filter plugin over libsqeazy_amd's C-ABI: the MI355X replacement of the
 * reference's filter (/root/reference/src/cpp/inc/sqeazy_h5_filter.hpp:28-227), same filter id (01307 octal = 711, the
 * reference's C++ octal literal at :212), same protocol:
 *   compress   cd_values[] hold the bytes of a sqy header (pipeline, voxel type, shape); the chunk is encoded with that
 *              pipeline and the stored chunk IS the blob SQY_PipelineEncode_* returns.  A chunk that already starts
 *              with a sqy header is stored as it is (:123-138).
 *   decompress the header in front of the chunk says everything (:44-104).
 * Built as sqeazy_amd/lib/libh5sqy_amd.so; point HDF5_PLUGIN_PATH at that directory, or register H5Z_SQY_AMD[0]
 * (H5PLget_plugin_info()) with H5Zregister.  Buffers are exchanged with HDF5 through malloc/free (the library frees the
 * buffer a filter returns with its own free; the reference uses new[]/delete[] there, which only works by accident). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "hdf5.h"
#include "H5PLextern.h"

#include "../../include/sqeazy_amd.h"

#define H5Z_FILTER_SQY 01307

static size_t H5Z_filter_sqy(unsigned flags, size_t cd_nelmts, const unsigned cd_values[], size_t nbytes, size_t* buf_size, void** buf)
{
    const char* in = (const char*)*buf;
    char* out = NULL;
    long outlen = 0;
    int ret = 1;

    if (flags & H5Z_FLAG_REVERSE) {
        long v = (long)nbytes;
        if (SQY_Decompressed_Sizeof(in, &v) == 0 && (v == 1 || v == 2)) {
            const long voxel = v;
            long len = (long)nbytes;
            if (SQY_Decompressed_Length(in, &len) == 0 && len > 0) {
                out = (char*)malloc((size_t)len);
                if (out) {
                    ret = voxel == 2 ? SQY_Decode_UI16(in, (long)nbytes, out, 0) : SQY_Decode_UI8(in, (long)nbytes, out, 0);
                    outlen = len;
                }
            }
        }
    } else {
        const char* hdr = (const char*)cd_values;
        const long hdr_bytes = (long)(cd_nelmts * sizeof(unsigned));
        long hs = (long)nbytes;
        if (SQY_Header_Size(in, &hs) == 0 && hs > 0) {
            /* the chunk is a blob already: store it unchanged */
            long payload = (long)nbytes;
            long rank = (long)nbytes;
            (void)rank;
            out = (char*)malloc(nbytes);
            if (out) { memcpy(out, in, nbytes); outlen = (long)nbytes; ret = 0; }
            (void)payload;
        } else {
            long plen = 0, rank = hdr_bytes, voxel = hdr_bytes;
            if (SQYAMD_Header_Pipeline(hdr, hdr_bytes, NULL, &plen) == 0 && SQY_Decompressed_NDims(hdr, &rank) == 0 && rank >= 1 && rank <= 16 &&
                SQY_Decompressed_Sizeof(hdr, &voxel) == 0 && (voxel == 1 || voxel == 2)) {
                char* pipeline = (char*)malloc((size_t)plen);
                long shape[16];
                shape[0] = hdr_bytes;
                if (pipeline && SQYAMD_Header_Pipeline(hdr, hdr_bytes, pipeline, &plen) == 0 && SQY_Decompressed_Shape(hdr, shape) == 0) {
                    size_t voxels = 1;
                    for (long i = 0; i < rank; ++i) voxels *= (size_t)shape[i];
                    long cap = (long)strlen(pipeline);
                    const int okc = voxel == 2 ? SQY_Pipeline_Max_Compressed_Length_3D_UI16(pipeline, shape, (unsigned)rank, &cap)
                                               : SQY_Pipeline_Max_Compressed_Length_3D_UI8(pipeline, shape, (unsigned)rank, &cap);
                    if (okc == 0 && voxels * (size_t)voxel == nbytes) {
                        out = (char*)malloc((size_t)cap);
                        if (out) {
                            /* the reference's filter encodes with a freshly built pipeline, i.e. n_threads = 1
                             * (inc/sqeazy_h5_filter.hpp:153-170, dynamic_pipeline.hpp:268): ONE block-linked LZ4 frame.
                             * Same bytes by default; SQY_H5_NTHREADS=0 selects the chunked layout (parallel on the GPU). */
                            const char* nt_env = getenv("SQY_H5_NTHREADS");
                            const int nt = nt_env ? atoi(nt_env) : 1;
                            ret = voxel == 2 ? SQY_PipelineEncode_UI16(pipeline, in, shape, (unsigned)rank, out, &outlen, nt)
                                             : SQY_PipelineEncode_UI8(pipeline, in, shape, (unsigned)rank, out, &outlen, nt);
                        }
                    } else {
                        fprintf(stderr, "[sqeazy]\t h5 filter: chunk of %zu bytes does not match the shape in cd_values\n", nbytes);
                    }
                }
                free(pipeline);
            }
        }
    }
    if (ret == 0 && out) {
        free(*buf);
        *buf = out;
        *buf_size = (size_t)outlen;
        return (size_t)outlen;
    }
    free(out);
    return 0;                                    /* HDF5 convention: 0 = failure */
}

const H5Z_class2_t H5Z_SQY_AMD[1] = {{
    H5Z_CLASS_T_VERS, (H5Z_filter_t)H5Z_FILTER_SQY, 1, 1,
    "HDF5 sqy filter (sqeazy_amd, MI355X); see https://github.com/sqeazy/sqeazy",
    NULL, NULL, (H5Z_func_t)H5Z_filter_sqy,
}};

__attribute__((visibility("default"))) H5PL_type_t H5PLget_plugin_type(void) { return H5PL_TYPE_FILTER; }
__attribute__((visibility("default"))) const void* H5PLget_plugin_info(void) { return H5Z_SQY_AMD; }
