/* Host side, part 6 of 6: inspection (what the binning kernels produced for the last pass), measurement (counters,
 * kernel timing), shutdown. */

/* --------------------------------------------------------------- inspection -- */

extern "C" int mi355r_inspect_pass(rz_pass_info *info) {
    if (!g.ready || !g.have_last_pass) return 0;
    graphics_flush();
    sync_stream();
    const PassParams &P = g.last_pass;
    uint32_t control[8];
    HIP_OK(hipMemcpy(control, P.control, sizeof control, hipMemcpyDeviceToHost));
    info->num_tris = P.num_tris;
    info->width = P.width;
    info->height = P.height;
    info->bin_edge = RZ_BIN;
    info->bins_x = P.bins_x;
    info->bins_y = P.bins_y;
    info->bin_entries = control[1];
    return 1;
}

extern "C" void mi355r_inspect_boxes(unsigned int *boxes, unsigned int *fans) {
    if (!g.have_last_pass) fatal("mi355r_inspect_boxes: no pass to look at");
    const PassParams &P = g.last_pass;
    const size_t T = (size_t)P.num_tris;
    std::vector<TriRec> recs(T);
    HIP_OK(hipMemcpy(boxes, P.bbox, T * sizeof(uint2), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(recs.data(), P.tri, T * sizeof(TriRec), hipMemcpyDeviceToHost));
    /* triangles of sky boxes that are kept on the pass's table have no box in the binning kernels' array: theirs
     * are put together from the table and their records */
    if (P.sky_draws > 0) {
        std::vector<uint32_t> table(RZ_SKY_TABLE), first_tri((size_t)P.num_draws + 1);
        HIP_OK(hipMemcpy(table.data(), P.sky_table, table.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(first_tri.data(), P.first_tri, first_tri.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        for (int k = 0; k < P.sky_draws; k++) {
            const int d = P.num_draws - P.sky_draws + k;
            for (uint32_t t = first_tri[d]; t < first_tri[d + 1]; t++) {
                const uint32_t *entry = &table[((size_t)k * RZ_SKY_FACES + (t - first_tri[d])) * RZ_MAX_FAN];
                bool any = false;
                for (int j = 0; j < RZ_MAX_FAN; j++) any = any || entry[j] != 0xffffffffu;
                if (!any) continue;
                const uint32_t fan = RZ_META_CLIPPED(recs[t].meta) ? RZ_META_FAN(recs[t].meta) : 1u;
                boxes[2 * t] = recs[t].bx;
                boxes[2 * t + 1] = (recs[t].by & 0x3fffu) | ((recs[t].by >> 16) << 14) | (fan << 28);
            }
        }
    }
    for (size_t t = 0; t < T; t++) {
        /* device packing (pack_bin_box): y0 | y1 << 14 | entries << 28 */
        const bool empty = (boxes[2 * t] & 0xffffu) > (boxes[2 * t] >> 16);
        const uint32_t y = boxes[2 * t + 1];
        fans[t] = empty ? 0u : y >> 28;
        boxes[2 * t + 1] = empty ? RZ_BBOX_EMPTY_X : ((y & 0x3fffu) | (((y >> 14) & 0x3fffu) << 16));
        if (!empty && fans[t] != (RZ_META_CLIPPED(recs[t].meta) ? RZ_META_FAN(recs[t].meta) : 1u))
            fatal("triangle %zu: packed entry count %u disagrees with its record", t, fans[t]);
    }
}

extern "C" void mi355r_inspect_bins(unsigned int *counts, unsigned int *entries) {
    if (!g.have_last_pass) fatal("mi355r_inspect_bins: no pass to look at");
    const PassParams &P = g.last_pass;
    const size_t T = (size_t)P.num_tris, bins = (size_t)P.bins_x * P.bins_y;
    std::vector<TriRec> recs(T);
    std::vector<uint2> boxes(T), ranges(bins);
    HIP_OK(hipMemcpy(recs.data(), P.tri, T * sizeof(TriRec), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(boxes.data(), P.bbox, T * sizeof(uint2), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(ranges.data(), P.bin_range, bins * sizeof(uint2), hipMemcpyDeviceToHost));
    /* fan block -> input triangle (a clipped triangle's record carries its block in ax) */
    std::unordered_map<uint32_t, uint32_t> owner;
    for (size_t t = 0; t < T; t++) {
        const bool empty = (boxes[t].x & 0xffffu) > (boxes[t].x >> 16);
        if (!empty && RZ_META_CLIPPED(recs[t].meta)) {
            uint32_t block;
            memcpy(&block, &recs[t].ax, sizeof block);
            owner[block] = (uint32_t)t;
        }
    }
    size_t out = 0;
    std::vector<BinEntry> list;
    for (size_t b = 0; b < bins; b++) {
        counts[b] = ranges[b].y;
        list.resize(ranges[b].y);
        if (ranges[b].y)
            HIP_OK(hipMemcpy(list.data(), P.bin_list + ranges[b].x, ranges[b].y * sizeof(BinEntry), hipMemcpyDeviceToHost));
        for (const BinEntry &e : list) {
            if (e.tri & 0x80000000u) {
                const uint32_t idx = e.tri & 0x7fffffffu;
                auto it = owner.find(idx / RZ_MAX_FAN);
                if (it == owner.end()) fatal("bin entry refers to fan block %u that no triangle owns", idx / RZ_MAX_FAN);
                entries[out++] = it->second;
                entries[out++] = idx % RZ_MAX_FAN;
            } else {
                entries[out++] = e.tri;
                entries[out++] = 0;
            }
        }
    }
}

/* -------------------------------------------------------------- measurement -- */

extern "C" void mi355r_enable_counters(int on) {
    ensure_ready();
    graphics_flush();
    g.counting = on != 0;
}

extern "C" void mi355r_reset_counters(void) {
    ensure_ready();
    graphics_flush();
    sync_stream();
    HIP_OK(hipMemset(g.d_counters, 0, 12 * sizeof(unsigned long long)));
    HIP_OK(hipDeviceSynchronize());
    memset(g.host_counters, 0, sizeof g.host_counters);
}

extern "C" void mi355r_get_counters(unsigned long long out[12]) {
    ensure_ready();
    graphics_flush();
    sync_stream();
    unsigned long long dev[12];
    HIP_OK(hipMemcpy(dev, g.d_counters, sizeof dev, hipMemcpyDeviceToHost));
    for (int i = 0; i < 8; i++) out[i] = g.host_counters[i];
    for (int i = 8; i < 12; i++) out[i] = 0;
    out[1] = dev[1];
    out[2] = dev[2];
    out[6] = dev[6];
    out[7] = dev[7];
    out[8] = dev[8];
    out[9] = dev[9];
}

/* Per-kernel device time, measured with HIP events on the pipeline's own
 * stream around every launch while enabled.  out[0..6]: summed milliseconds of
 * the vertex, setup+clip, bin, raster, skybox-shade, blinn-shade and pbr-shade
 * kernels (the shade slots stay zero for passes whose shading runs inside the
 * raster kernel); out[7]: launch sequences timed (each covers a batch of
 * passes); returns the number of passes. */
extern "C" void mi355r_enable_timing(int on) {
    ensure_ready();
    graphics_flush();
    sync_stream();
    g.timing = on != 0;
}

extern "C" unsigned long long mi355r_get_timing(double out[8], int reset) {
    ensure_ready();
    graphics_flush();
    sync_stream();
    for (size_t i = 0, b = 0; i + 7 < g.timing_events.size(); i += 8, b++) {
        for (int k = 0; k < 7; k++) {
            float ms = 0;
            HIP_OK(hipEventElapsedTime(&ms, g.timing_events[i + k], g.timing_events[i + k + 1]));
            g.kernel_ms[k] += ms;
        }
        g.kernel_launches += b < g.timing_passes.size() ? g.timing_passes[b] : 1;
        g.timed_sequences++;
    }
    g.timing_passes.clear();
    for (hipEvent_t e : g.timing_events) HIP_OK(hipEventDestroy(e));
    g.timing_events.clear();
    for (int k = 0; k < 7; k++) out[k] = g.kernel_ms[k];
    out[7] = (double)g.timed_sequences;
    unsigned long long n = g.kernel_launches;
    if (reset) { for (int k = 0; k < 7; k++) g.kernel_ms[k] = 0; g.kernel_launches = 0; g.timed_sequences = 0; }
    return n;
}

/* tests: force the first-time sizes of the clipper's block table and the bin lists (0 = the
 * library's own estimate), and count the passes that had to run again */
extern "C" void mi355r_set_scratch_estimate(unsigned long long fan_blocks, unsigned long long list_entries) {
    ensure_ready();
    graphics_flush();
    g.forced_fan_estimate = (size_t)fan_blocks;
    g.forced_list_estimate = (size_t)list_entries;
    if (!fan_blocks && !list_entries) g.fan_highwater = g.list_highwater = 0;
}

extern "C" unsigned long long mi355r_replayed_draws(void) { return g.replayed_draws; }

extern "C" const char *mi355r_last_raster_kernel(void) { return g.last_raster_kernel; }
#ifdef RZ_PROBE
extern "C" int mi355r_debug_probe(unsigned long long *out, int max_records) {
    HIP_OK(hipDeviceSynchronize());
    unsigned int n = 0;
    HIP_OK(hipMemcpyFromSymbol(&n, HIP_SYMBOL(rz_probe_count), sizeof n));
    n = std::min(n, (unsigned)RZ_PROBE_MAX);
    n = std::min(n, (unsigned)max_records);
    if (out && n) HIP_OK(hipMemcpyFromSymbol(out, HIP_SYMBOL(rz_probe_data), (size_t)n * RZ_PROBE_WORDS * sizeof(unsigned long long)));
    unsigned int zero = 0;
    HIP_OK(hipMemcpyToSymbol(HIP_SYMBOL(rz_probe_count), &zero, sizeof zero));
    return (int)n;
}
#endif

extern "C" int mi355r_inspect_sky_first(void) { return g.have_last_pass ? g.last_pass_sky_first : 0; }

extern "C" unsigned long long mi355r_passes_run_again(void) {
    ensure_ready();
    graphics_flush();
    sync_stream();
    return g.passes_run_again;
}

extern "C" void mi355r_shutdown(void) {
    if (!g.ready) return;
    graphics_flush();
    sync_stream();
}
