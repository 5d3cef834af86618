/*
 * heif_hipdec.h — C ABI of the MI355X-native HEIC decode path (libheif_amd/libheifhip.so).
 *
 * This is the drop-in boundary below libheif's decoder-plugin layer: plain C, pointers and sizes
 * only, no torch / C++ types.  Two groups of entry points:
 *
 *  1. hipdec_decoder_*  — one instance per coded image, the same life cycle libheif drives through
 *     heif_decoder_plugin (libheif/api/libheif/heif_plugin.h:85-169; call order in
 *     libheif/codecs/decoder.cc:355-563):
 *        new_decoder2 -> push_data2 (xN) -> flush_data -> decode_next_image2 -> free_decoder
 *     and it replaces what libheif/plugins/decoder_libde265.cc does with libde265
 *     (push: :322-368, decode: :386-457, plane hand-over: :97-171, VUI colour -> nclx: :426-449).
 *     hipdec_batch_* decodes many independent items (grid tiles, batches of stills) in one set of
 *     kernel launches — the device-side form of libheif/image-items/grid.cc:405-453.
 *
 *  2. hipdec_color_* — the fused colour stage over planes resident in HBM; each entry point
 *     restates one ColorConversionOperation of libheif/color-conversion (file:line at each
 *     declaration) bit-exactly (integer ops) or with identical float arithmetic (FMA contraction
 *     off).
 *
 * All `stride` arguments are in BYTES.  Device pointers are HIP device pointers of the device
 * selected with hipdec_init().  `stream` is a hipStream_t cast to void* (NULL = the library's
 * default stream).  Every function returns 0 on success or a negative hipdec_status; a
 * thread-local message is available from hipdec_last_error().
 */
#ifndef HEIF_HIPDEC_H
#define HEIF_HIPDEC_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define HIPDEC_API __attribute__((visibility("default")))
#else
#define HIPDEC_API
#endif

typedef enum hipdec_status {
  HIPDEC_OK = 0,
  HIPDEC_ERR_INVALID_ARGUMENT = -1,
  HIPDEC_ERR_END_OF_DATA = -2,       /* truncated NAL framing (heif_suberror_End_of_data)          */
  HIPDEC_ERR_BITSTREAM = -3,         /* malformed / non-conformant HEVC syntax                     */
  HIPDEC_ERR_UNSUPPORTED = -4,       /* valid HEVC outside the implemented tool set                */
  HIPDEC_ERR_LIMIT = -5,             /* security limit exceeded (max_image_size_pixels)            */
  HIPDEC_ERR_DEVICE = -6,            /* HIP runtime error / no device / kernel fault               */
  HIPDEC_ERR_NO_IMAGE = -7,          /* nothing decodable was pushed                               */
  HIPDEC_ERR_DECODE = -8,            /* device-side decode error (substream desynchronised ...)    */
  HIPDEC_ERR_MEMORY = -9             /* host allocation failed (heif_error_Memory_allocation_error) */
} hipdec_status;

/* ---- library ---------------------------------------------------------------------------------- */
HIPDEC_API int hipdec_init(int device_index);          /* idempotent; selects the device            */
HIPDEC_API void hipdec_shutdown(void);                  /* gives pools, streams and the resident-plane registry back; call it when no other thread is inside the
                                                          * library and no object of it is alive - a later call re-initialises the library by itself */
HIPDEC_API const char* hipdec_last_error(void);        /* thread-local, never NULL                  */
HIPDEC_API const char* hipdec_version(void);
HIPDEC_API int hipdec_device_count(void);

/* small device-memory helpers so that callers without a HIP toolchain (ctypes, cgo, JNI) can stage
 * buffers; real integrations pass their own device pointers */
HIPDEC_API void* hipdec_malloc(size_t bytes);
HIPDEC_API void hipdec_free(void* dptr);
HIPDEC_API int hipdec_memcpy_h2d(void* dst_dev, const void* src_host, size_t bytes);
HIPDEC_API int hipdec_memcpy_d2h(void* dst_host, const void* src_dev, size_t bytes);
HIPDEC_API int hipdec_memset(void* dst_dev, int value, size_t bytes);
HIPDEC_API int hipdec_stream_synchronize(void* stream);
/* additional HIP streams (hipStream_t as void*) so that independent batches overlap: the CABAC kernel keeps the
 * scalar pipes busy while another batch's reconstruction / filters / colour stage use the vector pipes and HBM */
HIPDEC_API void* hipdec_stream_create(void);
HIPDEC_API void hipdec_stream_destroy(void* stream);

/* ---- decoder ---------------------------------------------------------------------------------- */
typedef struct hipdec_decoder hipdec_decoder;

typedef struct hipdec_image_info {
  int width, height;              /* luma size after the conformance-window crop (== ispe)        */
  int chroma_format_idc;          /* 0 = 4:0:0, 1 = 4:2:0  (== heif_chroma numeric value)          */
  int chroma_width, chroma_height;
  int bit_depth_luma, bit_depth_chroma;
  /* VUI colour description as libde265 reports it (decoder_libde265.cc:426-449); H.265 Annex E
     defaults (2,2,2,limited) when absent */
  int colour_primaries, transfer_characteristics, matrix_coeffs, full_range_flag;
  int coded_width, coded_height;  /* pic_width/height_in_luma_samples                              */
  size_t bitstream_bytes;
  int num_substreams;             /* independent CABAC substreams (slices x tiles / WPP rows)      */
} hipdec_image_info;

/* new_decoder2 (heif_plugin.h:158; decoder_libde265.cc:175-214).  max_image_size_pixels == 0
 * means "no limit" (heif_security_limits.max_image_size_pixels, enforced before allocation as
 * decoder_libde265.cc:183-199 does). */
HIPDEC_API int hipdec_decoder_new(hipdec_decoder** out, int strict_decoding, uint64_t max_image_size_pixels);
HIPDEC_API void hipdec_decoder_free(hipdec_decoder* dec);
HIPDEC_API void hipdec_decoder_set_strict(hipdec_decoder* dec, int strict_decoding);

/* push_data2 (heif_plugin.h:160; decoder_libde265.cc:322-368): `data` is a concatenation of
 * [4-byte big-endian length][NAL unit without start code]; parameter sets first.  May be called
 * several times; the bytes are copied.  A push AFTER a decode starts the next picture (the samples of an intra-only image sequence,
 * libheif/sequences/track_visual.cc:200-280): the parameter sets seen so far stay active, so later samples may carry slice data only. */
HIPDEC_API int hipdec_decoder_push_data(hipdec_decoder* dec, const void* data, size_t size);

/* decode_next_image2 (heif_plugin.h:164; decoder_libde265.cc:386-457): parses the headers on the
 * host, runs the HIP decode pipeline and leaves the planes in HBM.  Returns HIPDEC_ERR_NO_IMAGE
 * when no picture was pushed (libheif sees "no image yet"). */
HIPDEC_API int hipdec_decoder_decode(hipdec_decoder* dec, hipdec_image_info* info);
/* The same with OUTPUT ORDER (de265_get_next_picture behind decoder_libde265.cc:402-419): decodes the pushed sample if one is pending, then
 * releases the next picture in output (POC) order when the bumping process of C.5.2.2 allows it - more pictures of the coded video sequence
 * are waiting than sps_max_num_reorder_pics, a new coded video sequence has started, or `flush` (the host's flush_data: end of the data).
 * *have = 0: no picture yet (libheif pushes the next sample).  Afterwards hipdec_decoder_read_plane* / _device_plane serve the released
 * picture.  Streams without B pictures come out in coding order, one picture per sample.  user_data: what hipdec_decoder_set_user_data
 * attached to the sample the picture was decoded from (push_data2's user_data, decoder_libde265.cc:360). */
HIPDEC_API void hipdec_decoder_set_user_data(hipdec_decoder* dec, uintptr_t user_data);
/* Look-ahead of sequence tracks: behind a decoder's first picture, hipdec_decoder_next_picture() decodes the pushed samples once `samples` of them
 * wait (or the host flushed): ONE launch set parses all of them (CABAC parsing needs nothing of a neighbour picture), the pixel stages follow picture
 * by picture in decoding order.  Until then it reports *have = 0 and libheif pushes the next sample (sequences/track_visual.cc:200-260).
 * 0 / 1: every sample is decoded at the poll behind its push.  Default 32 (environment: HIPDEC_SEQ_LOOKAHEAD; measured on 720p IPPP tracks: 19 fps without, 138 with 16, 205 with 32 samples); at most 64.  Stills are not affected:
 * the first picture of a decoder is always decoded at once. */
HIPDEC_API void hipdec_set_sequence_lookahead(int samples);
/* Pipelined chains: with `chains` > 1 a look-ahead chain is only ENQUEUED when its window is full, and its pictures are held back until `chains` of
 * them are in flight (or the host flushes) - libheif keeps pushing samples while no picture comes out (sequences/track_visual.cc:200-260), so the next
 * window fills while the chain runs and its CABAC launch (one WPP critical path, whatever the number of pictures) runs beside the pixel steps of the
 * chains in front of it.  The first picture of a chain in flight that goes out is preceded by the look at that chain's status; a chain that fails on
 * the device is undone together with the chains built on it and decoded again the plain way, so the sample that is to blame gets the error.
 * 1: every chain is waited for where it is launched.  Default 3 (environment: HIPDEC_SEQ_PIPELINE; at most 8; measured on 720p tracks, look-ahead 32: one IPPP
 * track 267 fps with 1, 550 with 3; 16 tracks side by side 1706 / 3093).  A host that wants a track's pictures with the least delay sets 1 (and a small look-ahead):
 * with D chains of L samples, up to D x L samples are pushed before the first of their pictures comes out.  Look-ahead 0 / 1 switches the pipeline off. */
HIPDEC_API void hipdec_set_sequence_pipeline(int chains);
/* statistics: chains that were left in flight when they were enqueued, and how often a failed one was undone */
HIPDEC_API void hipdec_decoder_pipeline_stats(uint64_t* chains_left_in_flight, uint64_t* rollbacks);
HIPDEC_API int hipdec_decoder_next_picture(hipdec_decoder* dec, int flush, hipdec_image_info* info, int* have, uintptr_t* user_data);
/* Concurrent hipdec_decoder_decode() calls (libheif decodes the tiles of a 'grid' item on worker threads,
 * libheif/image-items/grid.cc:405-453, one decoder instance per tile) are coalesced into shared launch sets; a
 * serial host never waits.  HIPDEC_COALESCE_WINDOW_US (default 2000, 0 = off) bounds the gathering time.
 * Counters since load: decode requests, launch sets issued, requests that shared a launch set with others. */
HIPDEC_API void hipdec_decoder_coalesce_stats(uint64_t* requests, uint64_t* launch_sets, uint64_t* shared_requests);
/* The same for sequence tracks decoded side by side (one decoder instance and one host thread per track, as libheif's Track_Visual objects,
 * libheif/sequences/track_visual.cc:200-280, are): the look-ahead chains (hipdec_set_sequence_lookahead) of instances that ask within
 * HIPDEC_CHAIN_WINDOW_US (default 20000; 0 = every chain on its own) of each other run as ONE launch set - step k of the set holds the k-th
 * dependency step of every track.  A leader only waits for instances that took part in a chain during the last second; a lone track never waits.
 * Counters since load: chains asked for, launch sets issued for them, launch sets that held more than one track's chain. */
HIPDEC_API void hipdec_decoder_chain_stats(uint64_t* chains, uint64_t* launch_sets, uint64_t* shared_launch_sets);

/* Host-only header probe: parses the parameter sets and slice segment headers of one pushed item
 * and fills `info` without touching the GPU (what libheif's own SPS pre-check does in
 * libheif/codecs/hevc_dec.cc:54-77, extended to the whole front end).  Same error codes as decode. */
HIPDEC_API int hipdec_probe(const void* data, size_t size, uint64_t max_image_size_pixels, hipdec_image_info* info);
/* For hosts that register plugins statically: heif_register_decoder_plugin(hipdec_get_decoder_plugin())
 * (libheif/api/libheif/heif_library.cc:70-81).  Returns a const heif_decoder_plugin*. */
HIPDEC_API const void* hipdec_get_decoder_plugin(void);

/* plane hand-over (decoder_libde265.cc:97-171): copies plane c (0 = Y, 1 = Cb, 2 = Cr) into a host
 * buffer with the caller's stride; samples are uint8 for bit depth 8, little-endian uint16
 * above.  A stride below the row length is refused (HIPDEC_ERR_INVALID_ARGUMENT: the last row would end past a buffer of
 * height x stride bytes); the same holds for every call that writes rows into a caller's buffer (hipdec_batch_read_plane,
 * hipdec_grid_read_plane, hipdec_batch_to_rgb, hipdec_color_convert, hipdec_grid_to_rgb). */
HIPDEC_API int hipdec_decoder_read_plane(hipdec_decoder* dec, int c, void* dst_host, size_t dst_stride);
/* the same, and remembers (host pointer -> device copy) so that hipdec_color_convert() on those very host planes skips the upload:
 * what the libheif plugin uses */
HIPDEC_API int hipdec_decoder_read_plane_tracked(hipdec_decoder* dec, int c, void* dst_host, size_t dst_stride);
/* Plane tracking on / off (default off; HIPDEC_TRACK_PLANES=1 in the environment turns it on).  While it is on,
 * hipdec_decoder_read_plane_tracked() records a hash over EVERY byte of the plane it copied out, so that a later
 * hipdec_color_convert() on the same host pointer can read the decoder's device copy instead of uploading — but only if the host
 * bytes are still exactly those (libheif edits planes in place between decode and conversion: image_item.cc:969 mirror_inplace).
 * An entry serves one conversion.  The plugin switches tracking on when the hosting libheif registers the HIP colour op
 * (INTEGRATION.md §4); the first hipdec_color_convert() call switches it on as well. */
HIPDEC_API void hipdec_set_plane_tracking(int on);
/* on = 2: for a host that ANNOUNCES every in-place edit of a handed-over plane with hipdec_forget_plane() before the colour conversion (the patched
 * libheif of libheif_amd/integration: image_item.cc:958-1004 mirror_inplace is its only one, and only as the fall-back of the device transform).  The
 * identity check then reads every 16th row instead of every byte (round 5: 38 ms of hashing per heif_decode_image() call under 256 threads). */
HIPDEC_API void hipdec_forget_plane(const void* host_plane);
/* drops every remembered (host pointer -> device copy) pair and with them the decode arenas they keep alive; hipdec_shutdown() and the
 * plugin's deinit_plugin() call it */
HIPDEC_API void hipdec_forget_resident_planes(void);
/* device-resident hand-over for callers that keep the colour stage on the GPU */
HIPDEC_API int hipdec_decoder_device_plane(hipdec_decoder* dec, int c, const void** dptr, size_t* stride);

/* Device arenas and pinned staging buffers of retired batches are parked for reuse up to this many bytes (default: a third of the
 * device's memory, single arenas up to two ninths of it - 96 / 64 GiB on an MI355X; pinned host buffers: an eighth of the host's RAM, at
 * most 24 GiB; HIPDEC_ARENA_CACHE_GIB overrides the device figure at hipdec_init): hipFree() synchronises the device, and the decoder
 * path builds one batch per launch set of concurrently decoding instances.  A host that wants the memory back lowers it; 0 empties the
 * cache. */
HIPDEC_API int hipdec_set_arena_cache_bytes(size_t bytes);

/* Two-stage pipeline across batches (default off): hipdec_batch_run() keeps the CABAC kernel on the caller's stream and queues residual /
 * reconstruction / deblock / SAO — and whatever follows for that batch: colour stage, packs — on a second stream of the device.  A host
 * that alternates two batches (two arenas) then has batch k+1's CABAC parse, which is issue- and dependency-bound, running beside batch k's
 * pixel stages.  hipdec_batch_status() / free wait for the batch as before. */
HIPDEC_API int hipdec_set_stage_overlap(int on);

/* Number of large batches the host keeps in flight at a time on separate streams (default 1).  The CABAC work pool of a
 * batch needs all its waves resident, so concurrent batches share the device's wave slots. */
HIPDEC_API int hipdec_set_concurrent_batches(int n);

/* Wave slots per SIMD (0 .. 4, default 0) the CABAC work pools leave free.  The pool's waves are resident for a whole launch set, so a host
 * that runs image-level kernels beside decoding - libheif with the colour / transformation hooks: the plugin sets 1 when it announces them -
 * keeps room for those, at 2 - 3 % of the parse rate; HIPDEC_RESERVED_WAVE_SLOTS overrides. */
HIPDEC_API int hipdec_set_reserved_wave_slots(int per_simd);

/* ---- batch decode (grid tiles / throughput mode) ------------------------------------------- */
typedef struct hipdec_batch hipdec_batch;
/* Parses n independent items (same framing as push_data; host worker threads for large batches) and uploads them; all
 * items must share chroma format and bit depth.  Large batches are staged in pinned memory and uploaded ASYNCHRONOUSLY on the
 * library's upload stream: the call returns when the host work is done, hipdec_batch_run() orders itself behind the copy, so
 * creating batch k+1 overlaps the kernels of batch k ("from compressed bytes in host memory").  `data` may be
 * released when the call returns.  hipdec_batch_run() is the device-resident hot path: inputs in HBM when it starts, decoded
 * planes in HBM when it (asynchronously) ends. */
HIPDEC_API int hipdec_batch_create(hipdec_batch** out, int n, const void* const* data, const size_t* sizes,
                                   uint64_t max_image_size_pixels);
/* The same for a stream of batches of one shape: the new batch takes over `recycle`'s arena (when it is large enough; `recycle`
 * may be NULL) and its upload is ordered behind whatever `recycle` still has in flight — decode, colour stage, packs.  One arena
 * then serves the whole stream while the host work of batch k+1 (header parsing, staging) overlaps the kernels of batch k.  After
 * the call `recycle` only answers hipdec_batch_status / timing queries and hipdec_batch_free: its planes are gone, so consume
 * them (hipdec_batch_to_rgb_all, hipdec_batch_pack_item) before creating the successor. */
HIPDEC_API int hipdec_batch_create_recycling(hipdec_batch** out, int n, const void* const* data, const size_t* sizes,
                                             uint64_t max_image_size_pixels, hipdec_batch* recycle);
HIPDEC_API void hipdec_batch_free(hipdec_batch* b);
HIPDEC_API int hipdec_batch_count(const hipdec_batch* b);
HIPDEC_API int hipdec_batch_info(const hipdec_batch* b, int i, hipdec_image_info* info);
HIPDEC_API int hipdec_batch_run(hipdec_batch* b, void* stream);      /* asynchronous                 */
HIPDEC_API int hipdec_batch_status(hipdec_batch* b);                 /* waits for THIS batch's work (not the stream); device errors */
HIPDEC_API int hipdec_batch_read_plane(hipdec_batch* b, int i, int c, void* dst_host, size_t dst_stride);
HIPDEC_API int hipdec_batch_device_plane(hipdec_batch* b, int i, int c, const void** dptr, size_t* stride);
/* Grid / multi-GPU hand-over (device-side form of HeifPixelImage::copy_image_to,
 * libheif/image/pixelimage.cc:1115-1172): hipdec_batch_pack_item writes item i's cropped planes tightly
 * (Y then Cb then Cr, row stride = width * bytes per sample) into a caller-owned device buffer, e.g. the
 * send buffer of an RCCL gather; hipdec_copy2d_d2d pastes a plane into a canvas at the caller's offset.
 * Both are asynchronous on `stream`. */
HIPDEC_API size_t hipdec_batch_item_packed_bytes(const hipdec_batch* b, int i);
HIPDEC_API int hipdec_batch_pack_item(hipdec_batch* b, int i, void* dst_dev, size_t dst_bytes, void* stream);
HIPDEC_API int hipdec_copy2d_d2d(void* dst_dev, size_t dst_stride, const void* src_dev, size_t src_stride, size_t width_bytes,
                                 size_t height, void* stream);
/* Fused colour stage over item i (planes -> interleaved RGB in HBM), see hipdec_color_* below;
 * out_chroma uses heif_chroma numeric values (10 = RGB, 11 = RGBA, 12/14 = RRGGBB BE/LE). */
HIPDEC_API int hipdec_batch_to_rgb(hipdec_batch* b, int i, int out_chroma, void* out_dev, size_t out_stride,
                                   void* stream);
/* The same for ALL items of the batch as one kernel launch (outs_dev[i] / out_strides[i] per item; every item must
 * select the same output layout, which out_chroma guarantees). */
HIPDEC_API int hipdec_batch_to_rgb_all(hipdec_batch* b, int out_chroma, void* const* outs_dev, const size_t* out_strides, void* stream);
/* hipdec_batch_run + hipdec_batch_to_rgb_all as one call.  For 8-bit 4:2:0 batches and interleaved RGB24 the colour conversion is FUSED into the
 * SAO kernel's store path (the planes are still written; the colour pass's re-read of them and its launch disappear); other cases run the two
 * steps one after the other.  Same pixels either way. */
HIPDEC_API int hipdec_batch_run_rgb(hipdec_batch* b, int out_chroma, void* const* outs_dev, const size_t* out_strides, void* stream);
/* per-kernel device time of the last run in microseconds (HIP events on the launch stream):
 * [0] CABAC parse, [1] reconstruction, [2] deblock, [3] SAO + crop, [4] total */
HIPDEC_API int hipdec_batch_last_timing_us(hipdec_batch* b, float out[5]);
/* keep the events of the last `slots` runs (run k records into slot k % slots) so that a benchmark can
 * average per-kernel device time over its whole timed region without synchronising between runs */
HIPDEC_API int hipdec_batch_timing_slots(hipdec_batch* b, int slots);
HIPDEC_API int hipdec_batch_slot_timing_us(hipdec_batch* b, int slot, float out[5]);
/* the same per kernel: [0] CABAC parse, [1] residual (dequantisation + inverse transforms), [2] intra reconstruction,
 * [3] deblock, [4] SAO + crop, [5] colour stage (hipdec_batch_to_rgb_all after that run; 0 if none), [6] decode total
 * (parse .. SAO), [7] reserved */
HIPDEC_API int hipdec_batch_slot_kernel_timing_us(hipdec_batch* b, int slot, float out[8]);
/* debug / test taps of item i after a run (device -> host): which = 0 pre-deblock, 1 post-deblock */
HIPDEC_API int hipdec_batch_read_tap(hipdec_batch* b, int i, int which, int c, void* dst_host, size_t dst_stride);
HIPDEC_API int hipdec_batch_read_maps(hipdec_batch* b, int i, uint8_t* log2_tb, uint8_t* log2_cb, uint8_t* intra_luma,
                                      uint8_t* intra_chroma, int8_t* qp_y, uint8_t* flags, size_t map_elems);

/* ---- grid images across the GPUs of one node ------------------------------------------------- */
/* Device-side form of ImageItem_Grid::decode_full_grid_image + decode_and_paste_tile_image (libheif/image-items/grid.cc:250-468,
 * :482-577; HeifPixelImage::copy_image_to, libheif/image/pixelimage.cc:1115-1172) in ONE process over several HIP devices: tile
 * t = row * cols + col (order of the 'dimg' references) is decoded by shard t mod n_devices on that shard's device, each decoded
 * plane is pasted with one strided device-to-device copy (xGMI peer access where available) at its position in the canvas on
 * devices[0], clipped to the output size; the colour conversion then runs once over the canvas.
 * devices: n_devices device indices (entries may repeat: several shards on one device); NULL = devices 0 .. n_devices-1,
 * n_devices 0 = all visible devices.  tile_data[t] / tile_sizes[t]: the tiles in grid order, push_data framing. */
typedef struct hipdec_grid hipdec_grid;
HIPDEC_API int hipdec_grid_create(hipdec_grid** out, int rows, int cols, int out_width, int out_height, const void* const* tile_data,
                                  const size_t* tile_sizes, const int* devices, int n_devices, uint64_t max_image_size_pixels);
HIPDEC_API void hipdec_grid_free(hipdec_grid* g);
/* info of the composed image (width / height = output size, coded size = tiled area, VUI colour description of tile 0) */
HIPDEC_API int hipdec_grid_info(const hipdec_grid* g, hipdec_image_info* info, int* n_shards);
HIPDEC_API int hipdec_grid_decode(hipdec_grid* g);                  /* asynchronous: decode + paste queued on every shard */
HIPDEC_API int hipdec_grid_wait(hipdec_grid* g);                    /* waits for all shards; device-side errors */
HIPDEC_API int hipdec_grid_canvas_plane(hipdec_grid* g, int c, const void** dptr, size_t* stride, int* device);
HIPDEC_API int hipdec_grid_read_plane(hipdec_grid* g, int c, void* dst_host, size_t dst_stride);
/* hipdec_grid_read_plane, and the host copy is remembered as device-resident (the canvas stays alive behind it until the entry is used):
 * what the ImageItem_Grid fast path of libheif_amd/integration/image_ops_hip.cc fills the composed HeifPixelImage with, so that the colour
 * conversion that follows (hipdec_color_convert) reads the canvas on the device */
HIPDEC_API int hipdec_grid_read_plane_tracked(hipdec_grid* g, int c, void* dst_host, size_t dst_stride);
/* the planner + fused colour stage over the canvas (hipdec_color_convert); out on the host, or on devices[0] */
HIPDEC_API int hipdec_grid_to_rgb(hipdec_grid* g, int out_chroma, int upsampling, int only_preferred, void* out, size_t out_stride,
                                  int out_on_device);

/* shards on the root device / shards that write the canvas through enabled peer access (xGMI) / shards whose copies the runtime stages
 * (hipDeviceCanAccessPeer said no, or hipDeviceEnablePeerAccess failed) */
HIPDEC_API int hipdec_grid_transport(const hipdec_grid* g, int* local_shards, int* peer_shards, int* staged_shards);

/* ---- grid images, SPMD form: one process per GPU, tiles gathered with RCCL over xGMI ----------------------------
 * The same partition as hipdec_grid_* (tile t -> rank t mod nranks; libheif/image-items/grid.cc:405-453, :482-577) for hosts that run one
 * process per GPU: every rank decodes its tiles on its own device, one grouped ncclSend / ncclRecv gather brings the packed tiles
 * (Y | Cb | Cr, 1.5 B/px for 8-bit 4:2:0) to rank 0, which pastes them (HeifPixelImage::copy_image_to, image/pixelimage.cc:1115-1172) and runs
 * the colour conversion once over the canvas.  create / decode are COLLECTIVE: every rank of the communicator calls them, with the same grid
 * geometry.  `comm` is an ncclComm_t (the host's own, or one from hipdec_rccl_comm_create); librccl is loaded at run time - without it these
 * entry points return HIPDEC_ERR_UNSUPPORTED. */
#define HIPDEC_RCCL_UNIQUE_ID_BYTES 128
HIPDEC_API int hipdec_rccl_available(void);
HIPDEC_API int hipdec_rccl_unique_id(void* id_out /* HIPDEC_RCCL_UNIQUE_ID_BYTES, rank 0; the host sends it to the other ranks */);
HIPDEC_API int hipdec_rccl_comm_create(void** comm_out, int nranks, int rank, const void* unique_id);   /* ncclCommInitRank on the hipdec_init device */
HIPDEC_API void hipdec_rccl_comm_destroy(void* comm);
typedef struct hipdec_grid_rccl hipdec_grid_rccl;
/* tile_data[t] / tile_sizes[t]: needed for the tiles this rank owns (t mod nranks == rank); other entries may be NULL / 0 */
HIPDEC_API int hipdec_grid_create_rccl(hipdec_grid_rccl** out, void* comm, int rank, int nranks, int rows, int cols, int out_width, int out_height,
                                       const void* const* tile_data, const size_t* tile_sizes, uint64_t max_image_size_pixels);
HIPDEC_API void hipdec_grid_rccl_free(hipdec_grid_rccl* g);
HIPDEC_API int hipdec_grid_rccl_decode(hipdec_grid_rccl* g);   /* asynchronous: decode + gather + paste queued on the rank's stream */
HIPDEC_API int hipdec_grid_rccl_wait(hipdec_grid_rccl* g);     /* this rank's work, then ONE 8-byte all-reduce per decode: every rank - rank 0 with the canvas first of all -
                                                                  * learns whether every shard decoded.  EVERY rank must call it (or to_rgb / read_plane, which do) after each decode -
                                                                  * ALSO a rank whose decode() returned an error (its peers are inside the all-reduce); it then returns that error */
/* rank 0 only (it owns the canvas): */
HIPDEC_API int hipdec_grid_rccl_canvas_plane(hipdec_grid_rccl* g, int c, const void** dptr, size_t* stride);
HIPDEC_API int hipdec_grid_rccl_read_plane(hipdec_grid_rccl* g, int c, void* dst_host, size_t dst_stride);
HIPDEC_API int hipdec_grid_rccl_to_rgb(hipdec_grid_rccl* g, int out_chroma, int upsampling, int only_preferred, void* out, size_t out_stride,
                                       int out_on_device);

/* ---- colour stage ---------------------------------------------------------------------------- */
typedef struct hipdec_nclx {
  int has_nclx;               /* 0: the image carries no nclx profile (reference uses its defaults) */
  int colour_primaries, transfer_characteristics, matrix_coefficients, full_range_flag;
} hipdec_nclx;

/* Op_YCbCr420_to_RGB24 / Op_YCbCr420_to_RGB32 (libheif/color-conversion/yuv2rgb.cc:345-426,
 * :481-562): 8-bit 4:2:0, 8.8 fixed point, nearest-neighbour chroma, alpha filled with 0xFF. */
HIPDEC_API int hipdec_color_420_to_rgb24(const void* y, size_t ys, const void* cb, size_t cbs, const void* cr, size_t crs,
                                         int w, int h, const hipdec_nclx* nclx, void* out, size_t out_stride,
                                         int with_alpha, void* stream);
/* Op_YCbCr_to_RGB<uint8_t/uint16_t> (yuv2rgb.cc:92-292): float32, NN chroma, planar R,G,B out with
 * the input's bit depth.  chroma: 1 = 4:2:0, 2 = 4:2:2, 3 = 4:4:4. */
HIPDEC_API int hipdec_color_ycbcr_to_rgb_planar(const void* y, size_t ys, const void* cb, size_t cbs, const void* cr,
                                                size_t crs, int w, int h, int bpp, int chroma, const hipdec_nclx* nclx,
                                                void* r, void* g, void* b, size_t out_stride, void* stream);
/* Op_YCbCr_to_RGB<uint8_t> followed by Op_RGB_to_RGB24_32 (rgb2rgb.cc:72-150) fused into one pass:
 * what libheif's planner runs for limited-range 8-bit input. */
HIPDEC_API int hipdec_color_ycbcr_to_rgb24_float(const void* y, size_t ys, const void* cb, size_t cbs, const void* cr,
                                                 size_t crs, int w, int h, int chroma, const hipdec_nclx* nclx,
                                                 void* out, size_t out_stride, int with_alpha, void* stream);
/* Op_YCbCr420_to_RRGGBBaa (yuv2rgb.cc:622-734): >8-bit 4:2:0 -> 16-bit interleaved BE or LE. */
HIPDEC_API int hipdec_color_420_to_rrggbb(const void* y, size_t ys, const void* cb, size_t cbs, const void* cr, size_t crs,
                                          int w, int h, int bpp, const hipdec_nclx* nclx, void* out, size_t out_stride,
                                          int little_endian, void* stream);
/* Op_YCbCr420_bilinear_to_YCbCr444<Pixel> (chroma_sampling.cc:501-724), one chroma plane;
 * (w, h) = luma size; reproduces the reference's border indexing. */
/* Op_mono_to_RGB24_32 (libheif/color-conversion/monochrome.cc): 8-bit monochrome plane -> RGB24 / RGBA32 (R = G = B = Y; alpha copied, or 0xFF) */
HIPDEC_API int hipdec_color_mono_to_rgb24(const void* y, size_t ys, const void* alpha, size_t alpha_stride, int w, int h, void* out,
                                          size_t out_stride, int with_alpha, void* stream);
/* > 8-bit planes (chroma 1 / 2 / 3) to 8-bit interleaved RGB(A) in one pass.  sdr_first = 1: Op_to_sdr_planes (hdr_sdr.cc:146-244) on the planes, then
 * Op_YCbCr420_to_RGB24 / _RGB32 (4:2:0 only); sdr_first = 0: Op_YCbCr_to_RGB<uint16_t>, Op_to_sdr_planes on R, G, B, Op_RGB_to_RGB24_32.  Which of
 * the two the reference's planner builds for a state: hipdec_color_plan. */
HIPDEC_API int hipdec_color_hdr_to_rgb24(const void* y, size_t ys, const void* cb, size_t cbs, const void* cr, size_t crs, int w, int h,
                                         int bpp, int chroma, const hipdec_nclx* nclx, void* out, size_t out_stride, int with_alpha,
                                         int sdr_first, void* stream);
/* Op_YCbCr_to_RGB<uint16_t> (libheif/color-conversion/yuv2rgb.cc:92-292) + Op_RGB_HDR_to_RRGGBBaa_BE (rgb2rgb.cc:470-560) [+ the endianness swap]:
 * > 8-bit planes of any chroma format (1 / 2 / 3) to interleaved RRGGBB at the input bit depth, one pass */
HIPDEC_API int hipdec_color_ycbcr_to_rrggbb_float(const void* y, size_t ys, const void* cb, size_t cbs, const void* cr, size_t crs,
                                                  int w, int h, int bpp, int chroma, const hipdec_nclx* nclx, void* out,
                                                  size_t out_stride, int little_endian, void* stream);
HIPDEC_API int hipdec_color_bilinear_420_to_444(const void* in, size_t is, int w, int h, int bpp, void* out, size_t os,
                                                void* stream);
/* Op_YCbCr422_bilinear_to_YCbCr444<Pixel> (chroma_sampling.cc:732-954), one chroma plane of ((w + 1) / 2) x h samples -> w x h. */
HIPDEC_API int hipdec_color_bilinear_422_to_444(const void* in, size_t is, int w, int h, int bpp, void* out, size_t os,
                                                void* stream);
/* Op_to_sdr_planes (hdr_sdr.cc:146-244): v >> (bits - 8), uint16 -> uint8. */
HIPDEC_API int hipdec_color_to_sdr(const void* in, size_t is, int w, int h, int bits, void* out, size_t os, void* stream);
/* Op_YCbCr420_to_RGB32 with a real alpha plane (yuv2rgb.cc:481-562; alpha copied as :552) when integer_op != 0, else the float
 * chain Op_YCbCr_to_RGB<uint8_t> + Op_RGB_to_RGB24_32 with the alpha plane interleaved (rgb2rgb.cc:72-150).  alpha NULL: 0xFF. */
HIPDEC_API int hipdec_color_420_to_rgba_alpha(const void* y, size_t ys, const void* cb, size_t cbs, const void* cr, size_t crs,
                                              int w, int h, const hipdec_nclx* nclx, const void* alpha, size_t alpha_stride,
                                              int integer_op, int chroma, void* out, size_t out_stride, void* stream);

/* ---- the colour boundary: planner (a8) + image-level conversion ------------------------------------------------------
 * What ColorConversionPipeline::construct_pipeline + convert_image do for a decoded HEIC image
 * (libheif/color-conversion/colorconversion.cc:279-623), behind ONE call.  The integration op registered in init_ops()
 * (libheif_amd/integration/colorconversion_hip.cc) forwards to hipdec_color_convert(). */
typedef enum hipdec_color_op {   /* the reference operations a plan is made of */
  HIPDEC_OP_TO_SDR = 1,                 /* Op_to_sdr_planes                  hdr_sdr.cc:146-244 */
  HIPDEC_OP_BILINEAR_420_TO_444 = 2,    /* Op_YCbCr420_bilinear_to_YCbCr444  chroma_sampling.cc:501-724 */
  HIPDEC_OP_420_TO_RGB24 = 3,           /* Op_YCbCr420_to_RGB24              yuv2rgb.cc:345-426 */
  HIPDEC_OP_420_TO_RGB32 = 4,           /* Op_YCbCr420_to_RGB32              yuv2rgb.cc:481-562 */
  HIPDEC_OP_YCBCR_TO_RGB = 5,           /* Op_YCbCr_to_RGB<Pixel>            yuv2rgb.cc:92-292 */
  HIPDEC_OP_RGB_TO_RGB24_32 = 6,        /* Op_RGB_to_RGB24_32                rgb2rgb.cc:72-150 (fused into the op before it) */
  HIPDEC_OP_420_TO_RRGGBB = 7,          /* Op_YCbCr420_to_RRGGBBaa           yuv2rgb.cc:622-734 */
  HIPDEC_OP_BILINEAR_422_TO_444 = 8,    /* Op_YCbCr422_bilinear_to_YCbCr444  chroma_sampling.cc:732-954 */
  HIPDEC_OP_RGB_HDR_TO_RRGGBB_BE = 9,   /* Op_RGB_HDR_to_RRGGBBaa_BE         rgb2rgb.cc (fused into the op before it) */
  HIPDEC_OP_SWAP_ENDIANNESS = 10,       /* Op_RRGGBBaa_swap_endianness       rgb2rgb.cc:647-764 (fused likewise) */
  HIPDEC_OP_MONO_TO_RGB24_32 = 11       /* Op_mono_to_RGB24_32               monochrome.cc (input chroma 0 = heif_chroma_monochrome) */
} hipdec_color_op;

typedef struct hipdec_color_image {
  int width, height;       /* luma size */
  int chroma;              /* heif_chroma of the planes: 1 = 4:2:0, 2 = 4:2:2, 3 = 4:4:4 */
  int bit_depth;
  const void* plane[4];    /* Y, Cb, Cr, alpha (NULL: none) */
  size_t stride[4];        /* bytes */
  int on_device;           /* 0: host pointers (libheif's HeifPixelImage planes), 1: device pointers */
} hipdec_color_image;

/* a8: the chain the reference's planner selects for (input state, output heif_chroma, chroma-upsampling options:
 * upsampling 1 = nearest neighbour, 2 = bilinear; only_preferred as heif_color_conversion_options).  HIPDEC_ERR_UNSUPPORTED for
 * states outside the HEIC hot path (the stock ops keep those). */
HIPDEC_API int hipdec_color_plan(int bit_depth, int chroma, int has_alpha, const hipdec_nclx* nclx, int out_chroma, int upsampling,
                                 int only_preferred, int ops[8], int* n_ops);
/* plan + execute on the GPU.  Host planes that a hipdec decoder has just handed to libheif (hipdec_decoder_read_plane_tracked)
 * are read from their device-resident copy, others are uploaded; `out` is a host buffer (or a device buffer, out_on_device). */
HIPDEC_API int hipdec_color_convert(const hipdec_color_image* in, const hipdec_nclx* nclx, int out_chroma, int upsampling,
                                    int only_preferred, void* out, size_t out_stride, int out_on_device);
/* ---- transformative item properties on the device: what ImageItem::decode_image applies to the decoded image on the host
 * (libheif/image-items/image_item.cc:949-1081: 'irot' :958-966, 'imir' :968-977, 'clap' :981-1010) ---------------------------------------- */
typedef enum hipdec_transform_op {
  HIPDEC_XF_ROTATE_CCW = 0,   /* args[0] = 90 / 180 / 270      HeifPixelImage::rotate_ccw      libheif/image/pixelimage.cc:1175-1333 */
  HIPDEC_XF_MIRROR = 1,       /* args[0] = heif_transform_mirror_direction (0 vertical, 1 horizontal)   mirror_inplace   :1337-1430 */
  HIPDEC_XF_CROP = 2          /* args = left, right, top, bottom (inclusive end points)        HeifPixelImage::crop            :1433-1530 */
} hipdec_transform_op;
/* Every plane of `in` (host or device pointers; host planes the decoder handed over are found device-resident) through the op, into the planes
 * `out` brings (same on_device convention; NULL where `in` has no plane); out->width / height / chroma / bit_depth are filled in.  Returns
 * HIPDEC_ERR_UNSUPPORTED where the reference converts the image to 4:4:4 first (odd size / offset of a subsampled image) and for odd crop
 * windows of subsampled images (half-covered chroma edge): the caller keeps the host path there. */
HIPDEC_API int hipdec_image_transform(const hipdec_color_image* in, int op, const int* args, hipdec_color_image* out);
/* the plane kernels (device pointers, `stream` a hipStream_t or NULL): ComponentStorage::rotate_ccw<T> / mirror_inplace<T> (pixelimage.cc:1305-1355)
 * and the per-plane copy of HeifPixelImage::crop; bytes_per_sample 1 or 2 */
HIPDEC_API int hipdec_plane_rotate_ccw(const void* in, size_t in_stride, int w, int h, int bytes_per_sample, int angle, void* out, size_t out_stride, void* stream);
HIPDEC_API int hipdec_plane_mirror(const void* in, size_t in_stride, int w, int h, int bytes_per_sample, int direction, void* out, size_t out_stride, void* stream);
HIPDEC_API int hipdec_plane_crop(const void* in, size_t in_stride, int w, int h, int bytes_per_sample, int left, int top, int out_w, int out_h, void* out,
                                 size_t out_stride, void* stream);

/* counters since load: images through hipdec_image_transform, grid canvases handed out by hipdec_grid_read_plane_tracked */
HIPDEC_API void hipdec_image_ops_stats(uint64_t* transforms, uint64_t* grid_canvases);
/* counters since load: conversions through hipdec_color_convert, input planes found device-resident, colour kernels launched */
HIPDEC_API void hipdec_color_boundary_stats(uint64_t* conversions, uint64_t* resident_planes, uint64_t* kernel_launches);
/* What the registry of handed-over planes holds right now: entries, and the device bytes they pin (every entry keeps its launch set's arena, a
 * transform result or a grid canvas alive; holders are counted once).  Bounded by age (HIPDEC_RESIDENT_TTL_MS, 3 s), by count, and by pinned bytes:
 * an eighth of the device's memory unless HIPDEC_RESIDENT_MAX_BYTES says otherwise (0: nothing is registered).  A device allocation that fails
 * empties the registry before its last retry. */
HIPDEC_API void hipdec_resident_plane_stats(uint64_t* entries, uint64_t* pinned_bytes);
/* Resident RGB: once a host has asked hipdec_color_convert() for interleaved RGB24 of 8-bit 4:2:0 planes in one op (what libheif's pipeline does
 * for heif_decode_image(.., heif_colorspace_RGB, heif_chroma_interleaved_RGB) through the integration op), the decoder's launch sets emit that RGB24
 * from the SAO kernel's store path (Op_YCbCr420_to_RGB24 / Op_YCbCr_to_RGB + Op_RGB_to_RGB24_32 fused, k_sao_rgb) and stage the rows to pinned host
 * memory beside the planes; a conversion whose planes are still the decoder's (every byte hashed) and whose colour description is the picture's VUI
 * is then a host copy.  Unfetched RGB spends the credit that requests earn, so a host that stops converting stops paying.  HIPDEC_NO_RESIDENT_RGB
 * switches it off.  images_produced: pictures decoded with RGB beside the planes; conversions_served: conversions answered from it. */
HIPDEC_API void hipdec_resident_rgb_stats(uint64_t* images_produced, uint64_t* conversions_served);

/* Op_to_hdr_planes (hdr_sdr.cc:25-109): 8-bit plane -> uint16 plane of out_bits (9..16): (v << (out_bits - 8)) | (v >> (16 - out_bits)). */
HIPDEC_API int hipdec_color_to_hdr(const void* in, size_t is, int w, int h, int out_bits, void* out, size_t os, void* stream);
/* Op_RRGGBBaa_swap_endianness (rgb2rgb.cc:647-764): interleaved RRGGBB (components 3) / RRGGBBAA (4) LE <-> BE. */
HIPDEC_API int hipdec_color_swap_endianness(const void* in, size_t is, int w, int h, int components, void* out, size_t os, void* stream);
/* PQ code values -> linear light (SMPTE ST 2084 / BT.2100 EOTF; 1.0 = 10000 cd/m2), float32 out with `components` values per pixel.
 * Input: 16-bit samples (interleaved RRGGBB[AA] rows or a plane, components = 1) of bit depth `bits`.  Not in the reference (libheif has
 * no transfer-function maths): BASELINE config 4's "PQ -> linear" stage; evaluated in fp64, tolerance against the published formula 1e-6 relative. */
HIPDEC_API int hipdec_color_pq_to_linear(const void* in, size_t is, int w, int h, int components, int bits, int big_endian, void* out, size_t os,
                                         void* stream);
/* The same for hybrid log-gamma code values (ARIB STD-B67 / BT.2100 HLG, transfer_characteristics 18): the inverse OETF per component, scene linear
 * light normalised to 1.0 (the display's OOTF is not applied).  Not in the reference either ("a real PQ / HLG EOTF stage"). */
HIPDEC_API int hipdec_color_hlg_to_linear(const void* in, size_t is, int w, int h, int components, int bits, int big_endian, void* out, size_t os,
                                          void* stream);
/* nclx.cc:143-173 get_YCbCr_to_RGB_coefficients: {r_cr, g_cb, g_cr, b_cb} */
HIPDEC_API void hipdec_color_coefficients(const hipdec_nclx* nclx, float out[4]);

#ifdef __cplusplus
}
#endif
#endif
