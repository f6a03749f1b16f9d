// Sanitizer harness of the JPEG host decoder (csrc/jpeg_host.cpp, plain C++: no HIP): built by tests/test_jpeg.py with
// g++ -fsanitize=address,undefined and run over a corpus of damaged files.  Every input is copied into a heap block of EXACTLY its
// size (so a read one byte past the file is a heap-buffer-overflow report, not a lucky zero), parsed, and - if the headers pass -
// entropy-decoded into an exactly sized coefficient buffer.  Prints one line per file; exits non-zero only through a sanitizer report.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "frp.h"
#include "jpeg_host.h"

int main(int argc, char** argv) {
    int decoded = 0, refused = 0;
    for (int i = 1; i < argc; ++i) {
        FILE* f = fopen(argv[i], "rb");
        if (!f) continue;
        fseek(f, 0, SEEK_END);
        const long n = ftell(f);
        fseek(f, 0, SEEK_SET);
        unsigned char* buf = (unsigned char*)malloc(n > 0 ? (size_t)n : 1);
        if (n > 0 && fread(buf, 1, (size_t)n, f) != (size_t)n) { fclose(f); free(buf); continue; }
        fclose(f);
        frp_jpeg_info info{};
        std::string err;
        int rc = frp::jpeg_info(buf, (size_t)n, &info, &err);
        if (rc == FRP_OK) {
            const size_t ce = frp::jpeg_coef_elems(info);
            if (ce > (size_t)64 << 20) { rc = FRP_ERR_INVALID; }          // (the pixel limit keeps real inputs far below this)
            else {
                int16_t* coef = (int16_t*)malloc(ce * 2 ? ce * 2 : 2);
                uint16_t q[192];
                rc = frp::jpeg_decode_coefficients(buf, (size_t)n, coef, ce, q, &info, &err);
                free(coef);
            }
        }
        {   // the plan of the device entropy decode (headers + restart-marker scan) over the same bytes
            frp::JpegDevicePlan plan;
            frp::JpegHuffTableDev tabs[6];
            std::string e2;
            const int rc2 = frp::jpeg_plan_device_decode(buf, (size_t)n, plan, tabs, &e2);
            if (rc2 == FRP_OK) {
                volatile unsigned sum = 0;
                for (size_t k = 0; k + 1 < plan.int_off.size(); ++k) sum += plan.scan[plan.int_off[k] < plan.scan_bytes ? plan.int_off[k] : 0];
                if (plan.scan_bytes) sum += plan.scan[plan.scan_bytes - 1];
            }
        }
        if (rc == FRP_OK) ++decoded; else ++refused;
        free(buf);
    }
    printf("decoded %d refused %d\n", decoded, refused);
    return 0;
}
