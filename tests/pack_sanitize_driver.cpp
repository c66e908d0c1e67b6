#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../raptor_amd/csrc/rq_kernels.hpp"   // test infrastructure: the product's host-side packers under ASan + UBSan
int main(int argc, char** argv) {
    std::vector<float> w(2084);
    FILE* f = fopen(argv[1], "rb"); if (!f || fread(w.data(), 4, 2084, f) != 2084) return 2; fclose(f);
    std::vector<float> a(rq::RQ_PACKED_FLOATS), b(rq::RQ_PACKED_BF16_FLOATS), c(rq::RQ_PACKED_F16X2_FLOATS), ls(rq::RQ_LOGSTD_FLOATS);
    rq::pack_policy(w.data(), a.data()); rq::pack_policy_bf16(w.data(), b.data()); rq::pack_policy_f16x2(w.data(), c.data());
    rq::pack_logstd_head(w.data(), w.data(), ls.data());
    double s = 0;
    for (int h1 : {16, 32, 64}) for (int h2 : {16, 32, 64}) for (int in : {13, 22}) {
        const size_t per = rq::teacher_param_count(in, h1, h2);
        std::vector<float> tw(per); for (size_t i = 0; i < per; ++i) tw[i] = 0.001f * (float)(i % 97) - 0.04f;
        std::vector<float> i32((size_t)rq::teacher_image_regs_f32(h1, h2) * 64), i16((size_t)rq::teacher_image_regs_bf16(h1, h2) * 64),
            isp((size_t)rq::teacher_image_regs_f16x2(h1, h2) * 64);
        rq::pack_teacher_f32(tw.data(), in, h1, h2, 1, 0, i32.data());
        rq::pack_teacher_bf16(tw.data(), in, h1, h2, 2, 2, i16.data());
        rq::pack_teacher_f16x2(tw.data(), in, h1, h2, 1, 2, isp.data());
        s += i32[5] + i16[7] + isp[9];
    }
    // the generic dense stack's streamed images (round 5): one, two and three hidden layers, both paddings, ragged widths
    for (int nh : {1, 2, 3}) for (int wmax : {48, 128}) for (int in : {7, 22}) {
        uint32_t widths[3] = {(uint32_t)wmax, 16u, (uint32_t)(wmax > 64 ? 112 : 32)};
        const int hp = wmax <= 64 ? 64 : 128;
        const size_t per = rq::teacher_layers_param_count(in, nh, widths);
        std::vector<float> tw(per); for (size_t i = 0; i < per; ++i) tw[i] = 0.002f * (float)(i % 89) - 0.05f;
        std::vector<float> img(rq::teacher_layers_image_floats(hp, nh));
        rq::pack_teacher_layers(tw.data(), in, nh, widths, hp, nh == 2 ? 2 : 1, nh == 3 ? 2 : 0, img.data());
        s += img[3] + img[img.size() - 1];
    }
    printf("pack under sanitizers ok %.3f %.3f\n", a[100] + b[100] + c[100], s);
    return 0;
}
