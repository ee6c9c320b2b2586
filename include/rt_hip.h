/*
 * rt_hip.h — C-ABI of librt_hip.so, the MI355X (gfx950) renderer for the `render_vectorized()`
 * hot path of AviadDahan/raytracer.
 *
 * The reference is pure Python/NumPy and has no FFI; each entry point below names the reference
 * function (file:line under the reference checkout) whose work it replaces.  The Python host code
 * (raytracer_amd/_lib.py) binds exactly these symbols with ctypes; INTEGRATION.md shows the stub a
 * maintainer of the reference would add.
 *
 * Conventions: plain C, no C++/torch types.  Every function returns an rt_status (0 = OK, < 0 =
 * error) and never throws or aborts; rt_last_error() returns a thread-local message for the last
 * failure on the calling thread.  All input arrays stay owned by the caller and are copied.
 */
#ifndef RT_HIP_H
#define RT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum rt_status {
  RT_OK = 0,
  RT_ERR_INVALID = -1,        /* bad argument (null pointer, negative size, unknown enum, ...) */
  RT_ERR_HIP = -2,            /* a HIP runtime call failed; message has hipGetErrorString */
  RT_ERR_NOMEM = -3,          /* device or host allocation failed */
  RT_ERR_QUEUE_OVERFLOW = -4, /* a bounce queue overflowed; blocking rt_render calls grow the queues and retry themselves */
  RT_ERR_NO_DEVICE = -5
} rt_status;

typedef enum rt_surface_type { RT_SPHERE = 0, RT_PLANE = 1, RT_BOX = 2 } rt_surface_type;
typedef enum rt_precision {
  RT_FP64 = 0, /* every operation in IEEE double, evaluation order of the reference */
  RT_FP32 = 1  /* fast mode: nearest-hit geometry in fp64, shadow/shading arithmetic in fp32 */
} rt_precision;
typedef enum rt_jitter_mode {
  RT_JITTER_PIXEL_KEYED = 0, /* u = hash(seed, pixel, depth, light, cell): pixels independent */
  RT_JITTER_NUMPY_STREAM = 1 /* the MT19937 stream np.random.seed(seed) feeds the reference, consumed in the
                                reference's order (ray_tracer.py:219-222); fp64, blocking calls only */
} rt_jitter_mode;
typedef enum rt_out_format { RT_OUT_F64 = 0, RT_OUT_F32 = 1, RT_OUT_U8 = 2 } rt_out_format;
enum { RT_FLAG_ASYNC = 1,  /* device output only: return after enqueueing; call rt_sync() */
       RT_FLAG_TIMING = 2, /* bracket every kernel with HIP events, fill the *_ms stats fields */
       RT_FLAG_SEQUENTIAL = 4, /* semantics of the reference's per-pixel recursive render() (ray_tracer.py:296-434)
                                 instead of render_vectorized: plane accepts t == 1e-6 (infinite_plane.py:20), unbounded
                                 shadow walk returning 0 below 1e-6 (:303-320), lights with intensity factor < 1e-6 skipped
                                 (:347-348), specular only for r.v > 0 (:356), background*transparency where the transmitted
                                 ray is not traced (:370-377).  Pixel-keyed jitter only; fp32 applies the last item only. */
       RT_FLAG_BRUTE_FORCE = 8 /* fp32 shade kernels test EVERY surface for EVERY shadow sample, like the reference's loop over
                                 all surfaces (ray_tracer.py:119-125): no candidate sets, no fully-lit / umbra shortcuts, no
                                 culling hierarchy.  Same image; exists so that the roofline fraction can be measured on a run
                                 that executes the algorithmic flop model's work (bench.py, roofline.brute_force). */ };

/*
 * Scene description, in scene-file (list) order — the order defines nearest-hit tie-breaking
 * (reference ray_tracer.py:119-125).  Mirrors the objects the reference's renderer reads:
 * surfaces (surfaces/sphere.py:5-8, infinite_plane.py:5-9, cube.py:5-12), the materials_array dict
 * (ray_tracer.py:454-460), Light (light.py:4-10), SceneSettings.background_color.
 */
typedef struct rt_scene_desc {
  int32_t n_surfaces;
  const int32_t* surf_type;     /* [n] rt_surface_type */
  const double* surf_data;      /* [n][8] sphere: cx cy cz r | plane: nx ny nz offset (n unit) | box: min xyz, max xyz */
  const int32_t* surf_material; /* [n] 0-based material index */
  int32_t n_materials;
  const double* materials;      /* [m][11] diffuse rgb, specular rgb, reflection rgb, shininess, transparency */
  int32_t n_lights;
  const double* lights;         /* [l][9] position xyz, color rgb, specular_intensity, shadow_intensity, radius */
  double background[3];
} rt_scene_desc;

/* Camera after Camera.setup() (reference camera.py:17-28). */
typedef struct rt_camera {
  double pos[3], forward[3], right[3], up[3];
  double screen_distance, screen_width, screen_height;
} rt_camera;

/*
 * One render call = the rows of a row set of one frame.  Row set: rows y in [0,height) with
 * ((y / block_rows) % n_parts) == part; n_parts <= 1 selects every row.  The output holds the
 * set's rows in ascending y, width*3 values each (the multi-GPU path renders one part per rank).
 */
typedef struct rt_frame {
  int32_t width, height;
  int32_t block_rows, part, n_parts;
  int32_t n_shadow;     /* SceneSettings.root_number_shadow_rays */
  int32_t max_depth;    /* SceneSettings.max_recursions */
  int32_t precision;    /* rt_precision */
  int32_t jitter_mode;  /* rt_jitter_mode */
  uint32_t jitter_seed;
  int32_t out_format;   /* rt_out_format; U8 = clip/x255/truncate of reference save_image, ray_tracer.py:606-608 */
  int32_t flags;        /* RT_FLAG_* */
} rt_frame;

#define RT_MAX_LEVELS 32
typedef struct rt_stats {
  int64_t rays;         /* R: radiance rays traced (reference's "total rays", ray_tracer.py:486,602) */
  int64_t hits;
  int64_t shadow_rays;  /* Sh = hits x lights x max(n,1)^2 */
  int64_t segments;     /* extra shadow-walk queries beyond the first per shadow ray (fp64 mode; -1 in fp32 mode) */
  int32_t batches;      /* non-empty levels (pixel-keyed) or DFS batches (numpy stream) */
  int32_t levels;
  int64_t level_rays[RT_MAX_LEVELS];
  int64_t level_hits[RT_MAX_LEVELS];
  float trace_ms, shade_ms, other_ms;   /* RT_FLAG_TIMING: summed kernel durations (HIP events) */
  int32_t trace_launches, shade_launches;
  float total_ms;                        /* RT_FLAG_TIMING: first kernel start -> last kernel end */
  int64_t retested;     /* fp32 mode: (hit, light) pairs whose shadow term was re-evaluated in fp64 because one of their
                           hit / miss decisions lay within the fp32 error bound (0 in fp64 mode) */
} rt_stats;

typedef struct rt_scene rt_scene;

const char* rt_last_error(void);
int rt_device_count(int* count);

/* Pack + upload a scene to `device` (type-sorted device records, original index kept).  Replaces
 * the per-call object walking of render_vectorized (ray_tracer.py:454-460, 518). */
int rt_scene_create(const rt_scene_desc* desc, int device, rt_scene** out);
void rt_scene_destroy(rt_scene* scene);

/* Use an external HIP stream (a hipStream_t, e.g. a torch.cuda.Stream's) for every later call on
 * this scene; NULL restores the scene's own stream, a private NON-BLOCKING stream (so NULL does not mean
 * the legacy default stream: work on it is not ordered with the caller's default-stream work). */
int rt_scene_set_stream(rt_scene* scene, void* hip_stream);
/* Wait for every RT_FLAG_ASYNC frame of the scene, fill their rt_stats.  Returns RT_ERR_QUEUE_OVERFLOW when one
 * of them — since the last rt_sync — overflowed its bounce queues (its image is incomplete; the queues have been
 * grown for the frames that follow: render it again). */
int rt_sync(rt_scene* scene);

/* Number of rows in a row set (host arithmetic only). */
int rt_rows_in_part(int32_t height, int32_t block_rows, int32_t part, int32_t n_parts, int32_t* rows);

/*
 * Render a row set.  Replaces render_vectorized (ray_tracer.py:437-604) including
 * Camera.generate_all_rays / generate_rays_for_rows (camera.py:46-104).
 * `out`: rows*width*3 elements of out_format, host memory (out_is_device = 0; the call blocks) or
 * device memory on the scene's device (out_is_device = 1; blocks unless RT_FLAG_ASYNC).
 * `stats` may be NULL; with RT_FLAG_ASYNC it is filled by the next rt_sync() on this scene
 * (the pointer must stay valid until then).
 */
int rt_render(rt_scene* scene, const rt_camera* camera, const rt_frame* frame, void* out, int out_is_device,
              rt_stats* stats);

/*
 * rt_render for callers that cannot allocate.  No counterpart in the reference: it exists for code under HIP graph
 * capture, tight serving loops and harnesses that forbid hipMalloc — rt_render grows a per-scene workspace on
 * demand, this variant never does.  Same kernels and results as rt_render with out_is_device = 1; everything a
 * frame needs beyond its input and output (counters, hit queue, two bounce-ray queues, a framebuffer when the
 * output is converted) is carved out of ONE static device array of RT_STATIC_SCRATCH_MIB MiB that each GPU the
 * library runs on holds until the process ends.  The call makes no hipMalloc / hipFree / hipHostMalloc and creates
 * no events (the scene handle and its pinned counter block were allocated by rt_scene_create).
 * Limits: rows*width <= RT_STATIC_MAX_PIXELS per call (RT_ERR_INVALID beyond; render larger frames as row sets);
 * pixel-keyed jitter; no RT_FLAG_TIMING; `out` is device memory.  The bounce queues take what the array has left
 * (>= 1.3 rays per pixel and level at the maximum size, more for smaller row sets); a frame that needs more fails
 * with RT_ERR_QUEUE_OVERFLOW — nothing is written out of bounds, and nothing is retried.  Blocks until the frame
 * is complete; with RT_FLAG_ASYNC (stats must be NULL) it only enqueues on the scene's stream — that is the variant
 * for HIP graph capture (the blocking one synchronises the stream, which a capture does not allow) — and the
 * overflow of such a frame is reported by the next rt_sync() on the scene (the flags of the LAST async static frame
 * before the sync: sync after each frame whose completeness matters).
 * All calls on one GPU share that GPU's array: calls of one process are serialised by the library, but an
 * RT_FLAG_ASYNC frame still in flight when the next call on the same GPU starts races with it (sync in between).
 */
#define RT_STATIC_SCRATCH_MIB 32
#define RT_STATIC_MAX_PIXELS 131072
int rt_render_static(rt_scene* scene, const rt_camera* camera, const rt_frame* frame, void* out_device, rt_stats* stats);

/* ONE frame rendered by several scene handles of this process, one per GPU (the same scene created with
 * rt_scene_create on each device): handle i renders the interleaved row blocks i, i + n, i + 2n, ... of `block_rows`
 * rows (frame->part / n_parts are ignored and set per handle), all handles at once, and every device copies its
 * blocks straight into `out` (one strided copy per device, over xGMI peer access when `out` is device memory on
 * another GPU, over PCIe when it is host memory).  No collective: the path has no exchange step (pixel-keyed jitter
 * makes rows independent).  Replaces the per-tile use of camera.py:75-104 (generate_rays_for_rows) + the caller's own
 * reassembly; SURVEY.md 8(b) names it as the multi-GPU boundary for callers without torch.distributed.
 * out: (height, width, 3) of frame->out_format, host memory (out_is_device = 0; pinned recommended) or device memory
 * of scenes[0]'s GPU (out_is_device = 1).  Pixel-keyed jitter only.  stats (may be NULL): summed over the handles.
 * Blocking; a handle whose bounce queues overflow renders its rows again by itself. */
int rt_render_multi(rt_scene* const* scenes, int n_scenes, const rt_camera* camera, const rt_frame* frame, void* out,
                    int out_is_device, rt_stats* stats);

/* Pinned (page-locked) host memory for output frames.  rt_render copies a host-output frame band by band while the
 * next band renders; into pinned memory those copies run asynchronously at PCIe speed.  No counterpart in the
 * reference, whose image is a NumPy array (ray_tracer.py:599): raytracer_amd.render wraps these buffers as arrays. */
int rt_host_alloc(size_t bytes, void** out);
void rt_host_free(void* p);

/* Test/utility exports (host pointers, fp64), one per reference helper:
 * rt_nearest_hit          == find_nearest_intersection_batch (ray_tracer.py:112-127)
 * rt_shadow_transmission  == compute_shadow_transmission_batch (ray_tracer.py:247-293)
 * rt_camera_rays          == Camera.generate_rays_for_rows (camera.py:75-104) */
int rt_nearest_hit(rt_scene* scene, const double* origins, const double* dirs, int64_t n_rays,
                   double* t_out, int32_t* idx_out, double* normals_out);
int rt_shadow_transmission(rt_scene* scene, const double* origins, const double* dirs, const double* max_dist,
                           int64_t n_rays, double* out);
int rt_camera_rays(rt_scene* scene, const rt_camera* camera, int32_t width, int32_t height, int32_t y0, int32_t y1,
                   double* origins_out, double* dirs_out);

#ifdef __cplusplus
}
#endif
#endif /* RT_HIP_H */
