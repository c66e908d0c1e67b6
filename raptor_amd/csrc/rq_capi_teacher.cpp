// rq_capi_teacher.cpp - the teacher bank: the distillation step of the reference (README.md:208-216: ~1000 MLP teachers queried on
// student-visited states), register-stationary family and dense stacks (rq_teacher.hip).
#include "rq_objects.hpp"

using namespace rqh;

extern "C" {

// ---------------------------------------------------------------------------- Teacher bank
RQ_API int rq_teacher_bank_create(rq_device* dev, const float* weights, uint32_t n_teachers, uint32_t in_dim, uint32_t h1,
                           uint32_t h2, int hidden_activation, int output_activation, rq_teacher_bank** out) {
    RQ_REQUIRE(dev && weights && out, RQ_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    RQ_REQUIRE(n_teachers > 0, RQ_ERR_INVALID_ARGUMENT, "n_teachers must be positive");
    RQ_REQUIRE(in_dim >= 1 && in_dim <= RQ_POLICY_INPUT_DIM, RQ_ERR_INVALID_ARGUMENT,
               "in_dim must be 1..22 (the recorded policy inputs)");
    auto ok_width = [](uint32_t h) { return h == 16 || h == 32 || h == 64; };
    RQ_REQUIRE(ok_width(h1) && ok_width(h2), RQ_ERR_INVALID_ARGUMENT, "hidden widths must be 16, 32 or 64");
    RQ_REQUIRE(hidden_activation == RQ_ACT_RELU || hidden_activation == RQ_ACT_TANH, RQ_ERR_INVALID_ARGUMENT,
               "hidden activation must be RQ_ACT_RELU or RQ_ACT_TANH");
    RQ_REQUIRE(output_activation == RQ_ACT_IDENTITY || output_activation == RQ_ACT_TANH, RQ_ERR_INVALID_ARGUMENT,
               "output activation must be RQ_ACT_IDENTITY or RQ_ACT_TANH");
    DeviceScope on_device(dev); int rc = on_device.rc; if (rc) return rc;
    rq_teacher_bank* b = new (std::nothrow) rq_teacher_bank();
    RQ_REQUIRE(b, RQ_ERR_OUT_OF_MEMORY, "host allocation failed");
    b->dev = dev; b->ordinal = dev->ordinal; b->n_teachers = n_teachers; b->in_dim = in_dim; b->h1 = h1; b->h2 = h2;
    b->act = hidden_activation; b->out_act = output_activation;
    const size_t per = rq::teacher_param_count((int)in_dim, (int)h1, (int)h2);
    const size_t f32_floats = (size_t)rq::teacher_image_regs_f32((int)h1, (int)h2) * 64;
    const size_t bf16_floats = (size_t)rq::teacher_image_regs_bf16((int)h1, (int)h2) * 64;
    const size_t split_floats = (size_t)rq::teacher_image_regs_f16x2((int)h1, (int)h2) * 64;
    std::vector<float> img32, img16, img_split;
    try {                                   // nothing throws across the boundary
        img32.resize(f32_floats * n_teachers);
        img16.resize(bf16_floats * n_teachers);
        img_split.resize(split_floats * n_teachers);
    } catch (const std::bad_alloc&) {
        delete b;
        return fail(RQ_ERR_OUT_OF_MEMORY, "rq_teacher_bank_create: host allocation failed");
    }
    for (uint32_t t = 0; t < n_teachers; ++t) {
        rq::pack_teacher_f32(weights + per * t, (int)in_dim, (int)h1, (int)h2, b->act, b->out_act, img32.data() + f32_floats * t);
        rq::pack_teacher_bf16(weights + per * t, (int)in_dim, (int)h1, (int)h2, b->act, b->out_act, img16.data() + bf16_floats * t);
        rq::pack_teacher_f16x2(weights + per * t, (int)in_dim, (int)h1, (int)h2, b->act, b->out_act, img_split.data() + split_floats * t);
    }
    hipError_t e1 = hipMalloc(&b->images_f32, img32.size() * sizeof(float));
    hipError_t e2 = hipMalloc(&b->images_bf16, img16.size() * sizeof(float));
    if (e1 == hipSuccess) e1 = hipMemcpy(b->images_f32, img32.data(), img32.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e2 == hipSuccess) e2 = hipMemcpy(b->images_bf16, img16.data(), img16.size() * sizeof(float), hipMemcpyHostToDevice);
    hipError_t e3 = hipMalloc(&b->images_f16x2, img_split.size() * sizeof(float));
    if (e3 == hipSuccess) e3 = hipMemcpy(b->images_f16x2, img_split.data(), img_split.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) {
        rq_teacher_bank_destroy(b);
        return fail(RQ_ERR_OUT_OF_MEMORY, "rq_teacher_bank_create: device allocation or upload failed");
    }
    *out = b;
    return RQ_OK;
}

RQ_API int rq_teacher_bank_create_layers(rq_device* dev, const float* weights, uint32_t n_teachers, uint32_t in_dim, uint32_t n_hidden,
                                  const uint32_t* widths, int hidden_activation, int output_activation, rq_teacher_bank** out) {
    RQ_REQUIRE(dev && weights && widths && out, RQ_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    RQ_REQUIRE(n_hidden >= 1 && n_hidden <= 3, RQ_ERR_INVALID_ARGUMENT, "a teacher has one, two or three hidden layers");
    auto fast_width = [](uint32_t h) { return h == 16 || h == 32 || h == 64; };
    if (n_hidden == 2 && fast_width(widths[0]) && fast_width(widths[1]))      // the register-stationary family (three precisions)
        return rq_teacher_bank_create(dev, weights, n_teachers, in_dim, widths[0], widths[1], hidden_activation, output_activation, out);
    RQ_REQUIRE(n_teachers > 0, RQ_ERR_INVALID_ARGUMENT, "n_teachers must be positive");
    RQ_REQUIRE(in_dim >= 1 && in_dim <= RQ_POLICY_INPUT_DIM, RQ_ERR_INVALID_ARGUMENT,
               "in_dim must be 1..22 (the recorded policy inputs)");
    uint32_t widest = 0;
    for (uint32_t l = 0; l < n_hidden; ++l) {
        RQ_REQUIRE(widths[l] >= 16 && widths[l] <= 128 && widths[l] % 16 == 0, RQ_ERR_INVALID_ARGUMENT,
                   "hidden widths must be multiples of 16 from 16 to 128");
        widest = widths[l] > widest ? widths[l] : widest;
    }
    RQ_REQUIRE(hidden_activation == RQ_ACT_RELU || hidden_activation == RQ_ACT_TANH, RQ_ERR_INVALID_ARGUMENT,
               "hidden activation must be RQ_ACT_RELU or RQ_ACT_TANH");
    RQ_REQUIRE(output_activation == RQ_ACT_IDENTITY || output_activation == RQ_ACT_TANH, RQ_ERR_INVALID_ARGUMENT,
               "output activation must be RQ_ACT_IDENTITY or RQ_ACT_TANH");
    DeviceScope on_device(dev); int rc = on_device.rc; if (rc) return rc;
    rq_teacher_bank* b = new (std::nothrow) rq_teacher_bank();
    RQ_REQUIRE(b, RQ_ERR_OUT_OF_MEMORY, "host allocation failed");
    b->dev = dev; b->ordinal = dev->ordinal; b->n_teachers = n_teachers; b->in_dim = in_dim;
    b->act = hidden_activation; b->out_act = output_activation;
    b->layers = true; b->n_hidden = n_hidden; b->hp = widest <= 64 ? 64u : 128u;
    for (uint32_t l = 0; l < n_hidden; ++l) b->widths[l] = widths[l];
    b->h1 = widths[0]; b->h2 = n_hidden > 1 ? widths[1] : 0;
    const size_t per = rq::teacher_layers_param_count((int)in_dim, (int)n_hidden, widths);
    const size_t floats = rq::teacher_layers_image_floats((int)b->hp, (int)n_hidden);
    std::vector<float> img;
    try {                                   // nothing throws across the boundary
        img.resize(floats * n_teachers);
    } catch (const std::bad_alloc&) {
        delete b;
        return fail(RQ_ERR_OUT_OF_MEMORY, "rq_teacher_bank_create_layers: host allocation failed");
    }
    for (uint32_t t = 0; t < n_teachers; ++t)
        rq::pack_teacher_layers(weights + per * t, (int)in_dim, (int)n_hidden, widths, (int)b->hp, b->act, b->out_act, img.data() + floats * t);
    hipError_t e = hipMalloc(&b->images_layers, img.size() * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(b->images_layers, img.data(), img.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        rq_teacher_bank_destroy(b);
        return fail(RQ_ERR_OUT_OF_MEMORY, "rq_teacher_bank_create_layers: device allocation or upload failed");
    }
    *out = b;
    return RQ_OK;
}

RQ_API int rq_teacher_bank_destroy(rq_teacher_bank* bank) {
    if (!bank) return RQ_OK;
    DeviceScope on_device(bank->ordinal);
    if (bank->images_layers) (void)hipFree(bank->images_layers);
    if (bank->images_f32) (void)hipFree(bank->images_f32);
    if (bank->images_bf16) (void)hipFree(bank->images_bf16);
    if (bank->images_f16x2) (void)hipFree(bank->images_f16x2);
    if (bank->tiles) (void)hipFree(bank->tiles);
    delete bank;
    return RQ_OK;
}

RQ_API int rq_teacher_bank_set_precision(rq_teacher_bank* bank, int precision) {
    RQ_REQUIRE(bank, RQ_ERR_INVALID_ARGUMENT, "null argument");
    RQ_REQUIRE(precision == RQ_POLICY_FP32 || precision == RQ_POLICY_BF16_MFMA || precision == RQ_POLICY_F16X2_MFMA,
               RQ_ERR_INVALID_ARGUMENT, "unknown precision");
    RQ_REQUIRE(!bank->layers || precision == RQ_POLICY_FP32, RQ_ERR_INVALID_ARGUMENT,
               "a bank outside the two-hidden-layer {16, 32, 64} family is evaluated in fp32 only");
    bank->precision = precision;
    return RQ_OK;
}

RQ_API int rq_trajectory_relabel_teachers(rq_trajectory* t, rq_teacher_bank* bank, const uint32_t* teacher_id, float* action_out,
                                   int overwrite) {
    RQ_REQUIRE(t && bank && teacher_id, RQ_ERR_INVALID_ARGUMENT, "null argument");
    rq_env* env = t->env;
    rq_device* dev = env->dev;
    RQ_REQUIRE(bank->dev == dev, RQ_ERR_SHAPE_MISMATCH, "teacher bank lives on another device");
    if (t->length == 0) return RQ_OK;
    const uint32_t n = env->n;
    // group the envs by teacher: a tile = up to 16 envs of ONE teacher (counting sort over the teacher ids, env
    // order kept inside a teacher, so sorted inputs give contiguous tiles and coalesced rows)
    for (uint32_t i = 0; i < n; ++i)
        RQ_REQUIRE(teacher_id[i] < bank->n_teachers, RQ_ERR_INVALID_ARGUMENT, "teacher id out of range");
    // register-stationary family: tile_teacher [n_tiles] | tile_env [n_tiles][16] (a tile = up to 16 envs of ONE teacher);
    // dense stacks (round 6): teacher_start [n_teachers + 1] | sorted_env [n] - the kernel forms its 16-wide tiles out of (env, step) pairs
    std::vector<uint32_t> host;
    uint32_t n_tiles = 0;
    if (bank->layers)
        RQ_REQUIRE((uint64_t)n * t->length < (1ull << 32), RQ_ERR_INVALID_ARGUMENT, "envs x steps must stay below 2^32 for a dense-stack bank");
    try {                                   // nothing throws across the boundary
        std::vector<uint32_t> count(bank->n_teachers, 0), start(bank->n_teachers, 0), filled(bank->n_teachers, 0);
        for (uint32_t i = 0; i < n; ++i) ++count[teacher_id[i]];
        if (bank->layers) {
            host.assign((size_t)bank->n_teachers + 1 + n, 0u);
            uint32_t at = 0;
            for (uint32_t k = 0; k < bank->n_teachers; ++k) { host[k] = start[k] = at; at += count[k]; }
            host[bank->n_teachers] = at;
            for (uint32_t i = 0; i < n; ++i) {
                const uint32_t k = teacher_id[i];
                host[(size_t)bank->n_teachers + 1 + start[k] + filled[k]++] = i;
            }
        } else {
            for (uint32_t k = 0; k < bank->n_teachers; ++k) { start[k] = n_tiles; n_tiles += (count[k] + 15u) / 16u; }
            host.assign((size_t)n_tiles * 17, 0xFFFFFFFFu);
            for (uint32_t i = 0; i < n; ++i) {
                const uint32_t k = teacher_id[i], pos = filled[k]++;
                const uint32_t tile = start[k] + pos / 16u;
                host[tile] = k;
                host[(size_t)n_tiles + (size_t)tile * 16 + pos % 16u] = i;
            }
        }
    } catch (const std::bad_alloc&) {
        return fail(RQ_ERR_OUT_OF_MEMORY, "rq_trajectory_relabel_teachers: host allocation failed");
    }
    DeviceScope on_device(dev); int rc = on_device.rc; if (rc) return rc;
    if (bank->tile_words < host.size()) {
        RQ_HIP(hipStreamSynchronize(dev->stream));
        if (bank->tiles) { RQ_HIP(hipFree(bank->tiles)); bank->tiles = nullptr; bank->tile_words = 0; }
        RQ_HIP(hipMalloc(&bank->tiles, host.size() * sizeof(uint32_t)));
        bank->tile_words = host.size();
    }
    RQ_HIP(hipMemcpyAsync(bank->tiles, host.data(), host.size() * sizeof(uint32_t), hipMemcpyHostToDevice, dev->stream));
    RQ_HIP(hipStreamSynchronize(dev->stream));                // `host` is pageable and about to go out of scope
    const size_t act_bytes = (size_t)t->length * RQ_ACTION_DIM * env->ld * sizeof(float);
    float* d_act = t->act;
    if (!overwrite) {
        if (dev->rows2_bytes < act_bytes) {
            if (dev->rows2) { RQ_HIP(hipFree(dev->rows2)); dev->rows2 = nullptr; dev->rows2_bytes = 0; }
            RQ_HIP(hipMalloc(&dev->rows2, act_bytes));
            dev->rows2_bytes = act_bytes;
        }
        d_act = dev->rows2;
    }
    const float* images = bank->precision == RQ_POLICY_BF16_MFMA ? bank->images_bf16
                        : bank->precision == RQ_POLICY_F16X2_MFMA ? bank->images_f16x2 : bank->images_f32;
    if (bank->layers)
        RQ_HIP(rq::launch_teacher_relabel_layers(dev->stream, bank->n_teachers, n, env->ld, t->length, bank->in_dim, bank->n_hidden, bank->hp,
                                                 bank->act, bank->out_act, bank->images_layers, bank->tiles, bank->tiles + bank->n_teachers + 1,
                                                 t->obs, d_act));
    else
    RQ_HIP(rq::launch_teacher_relabel(dev->stream, n_tiles, env->ld, t->length, bank->in_dim, bank->h1, bank->h2, bank->act,
                                      bank->out_act, bank->precision, images, bank->tiles, bank->tiles + n_tiles, t->obs,
                                      d_act));
    if (action_out) return traj_block_to_host(dev, d_act, t->length, env->n, env->ld, RQ_ACTION_DIM, action_out);
    return RQ_OK;
}

}  // extern "C"
