// CPU AddressSanitizer harness for the host-only JPEG decoder (sanitizers run on the CPU build only).
#include <cstdio>
#include <cstdint>
#include <vector>
struct icl_ctx;
int icl_jpeg_decode(icl_ctx *ctx, const uint8_t *data, size_t len, const char *path, std::vector<uint8_t> &rgb, int &W, int &H);
int main(int argc, char **argv)
{
    int ok = 0, bad = 0;
    for (int i = 1; i < argc; ++i) {
        FILE *f = fopen(argv[i], "rb");
        if (!f) continue;
        std::vector<uint8_t> d;
        int c;
        while ((c = fgetc(f)) != EOF) d.push_back((uint8_t)c);
        fclose(f);
        std::vector<uint8_t> rgb;
        int w, h;
        (icl_jpeg_decode(nullptr, d.data(), d.size(), argv[i], rgb, w, h) == 0 ? ok : bad)++;
    }
    printf("decoded %d, rejected %d\n", ok, bad);
}
