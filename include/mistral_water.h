/*
 * mistral_water.h -- C ABI of libmistral_water.so, the MI355X (gfx950) ocean heightfield synthesiser.
 *
 * The reference (AlphaMistral/Mistral-Water, a Unity C# project) has NO native plugin / P/Invoke
 * interface: the boundary is cut here at the seam between the MonoBehaviour drivers
 * and the private numerical methods they call on themselves.  Every entry point cites the reference
 * code it replaces (S/ = Assets/Mistral Water/Scripts/, F/ = .../Shaders/FFT/, W/ = .../Shaders/).
 *
 * Conventions
 *  - plain C, no C++ types, no callbacks; status codes only (never exceptions / aborts);
 *    mw_last_error() returns a thread-local description of the last failure.
 *  - arrays use the reference's layouts: Vector2 = 2 x f32, Vector3 = 3 x f32, Color = 4 x f32,
 *    grid index idx = i*N + j with i along x and j along z (S/FFTMesh.cs:110).
 *  - "host" entry points take host pointers and are synchronous (kernels + D2H done on return);
 *    "_device" entry points take device pointers, enqueue on the handle's stream and return at once.
 *  - one mw_ocean must not be used from two threads at once; distinct handles may be.
 *  - the library is HIP-only: there is NO CPU fallback.  mw_ocean_create fails with MW_EDEVICE when
 *    no gfx950 device is usable.
 */
#ifndef MISTRAL_WATER_H
#define MISTRAL_WATER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MW_ABI_VERSION 4

typedef struct mw_ocean mw_ocean; /* opaque; one per FFTMesh / OceanRenderer instance */

typedef enum {
    MW_OK = 0,
    MW_EINVAL = 1,           /* bad argument / NULL pointer / unsupported resolution */
    MW_ENOTPOW2 = 2,         /* OceanRenderer semantics needs a power-of-two texture size (S/OceanRenderer.cs:231) */
    MW_ENOTCOMMENSURATE = 3, /* reserved: (unit_width != length/N is served by the direct-sum kernel instead) */
    MW_ENOMEM = 4,
    MW_EDEVICE = 5,          /* no usable HIP device / HIP runtime error */
    MW_ESTATE = 6            /* call order violation (e.g. evaluate before a spectrum exists) */
} mw_status;

typedef enum {
    MW_SEM_FFTMESH = 0,       /* S/FFTMesh.cs: centred k, quantised dispersion, closed-form time, spectral normals */
    MW_SEM_OCEANRENDERER = 1  /* S/OceanRenderer.cs + the F/ shaders: FFT-order k, capillary dispersion, iterated phase */
} mw_semantics;

/* Public Inspector fields of S/FFTMesh.cs:9-23 and S/OceanRenderer.cs:10-19, plus the constants the
 * reference hard-codes (gravity = G = 9.81f, S/FFTMesh.cs:52, F/FFTCommon.cginc:9). */
typedef struct {
    int32_t resolution; /* FFTMesh: grid is resolution^2.  OceanRenderer: textures are (8*resolution)^2 (S/OceanRenderer.cs:136) */
    float unit_width;   /* unitWidth */
    float length;       /* length */
    float wind_x, wind_y; /* wind */
    float amplitude;    /* amplitude (OceanRenderer divides by 10000 itself, S/OceanRenderer.cs:149) */
    float choppiness;   /* choppiness */
    float gravity;      /* 9.81f in the reference */
    float t_division;   /* FFTMesh.tDivision  (S/FFTMesh.cs:70) */
    float mult;         /* OceanRenderer.mult (S/OceanRenderer.cs:223) */
    uint64_t seed;      /* seed of the library's documented counter RNG (the reference never seeds Unity's) */
    int32_t semantics;  /* mw_semantics */
    int32_t device;     /* HIP device ordinal */
} mw_params;

/* flags of mw_ocean_evaluate_device */
#define MW_OUT_WHITE_SCALAR 0u /* whitecap written as 1 float per vertex (the canonical 28 B/point output) */
#define MW_OUT_COLOR_RGBA 1u   /* whitecap replicated into a Unity Color (S/FFTMesh.cs:274), 16 B per vertex */

int32_t mw_abi_version(void);
/* Identity of this build of the library: "<16 hex digits of a hash over the kernel sources and compile flags> <flags tag>".
 * bench.py prints it and the committed rocprofv3 PMC summaries carry it, so a counter file is only ever quoted next to
 * numbers of the build it was measured on.                                                                          */
const char* mw_build_id(void);
const char* mw_last_error(void);
int32_t mw_device_count(void);
void mw_params_default(mw_params* p, int32_t semantics); /* Inspector defaults of the two MonoBehaviours */

/* ---- lifecycle --------------------------------------------------------------------------------
 * mw_ocean_create = SetParams + GenerateMesh's spectrum fill (S/FFTMesh.cs:90-99,114-116;
 * S/OceanRenderer.cs:116-170,209-214): allocates device state and generates h0 / h0conj on the GPU
 * from params->seed (Phillips, S/FFTMesh.cs:149-166; Box-Muller htilde0, :168-176).                */
mw_status mw_ocean_create(const mw_params* params, mw_ocean** out);
void mw_ocean_destroy(mw_ocean* o);
/* OceanRenderer semantics, `ntiles` independent oceans in ONE handle: tile k is the ocean of `params` with seed
 * params->seed + k.  One 1024^2 frame is three launches of a few hundred workgroups -- latency-, not bandwidth-bound -- and
 * the stateful phase (F/FFTCommon.cginc:101-104) forbids batching in time, so the tile axis is what fills the device: every
 * GenerateTexture() of the handle advances all tiles in the same three launches.  Every OceanRenderer entry point of a
 * batched handle takes / returns arrays with a leading tile axis ([ntiles][M*M*...], [ntiles][resolution^2*...] for
 * mw_ocean_displace_mesh); the rest mesh is one mesh.  Per-tile results are identical to single handles of seed + k.     */
mw_status mw_ocean_create_batch(const mw_params* params, int32_t ntiles, mw_ocean** out);
int32_t mw_ocean_batch_size(const mw_ocean* o);

/* Run all subsequent work of this handle on an existing hipStream_t (e.g. torch's current stream).  The argument means
 * what it says: NULL is HIP's legacy default stream (what torch.cuda.current_stream().cuda_stream is by default), as in
 * the pond entry points.  A fresh handle runs on its own private non-blocking stream; mw_ocean_use_own_stream returns to
 * it.  Work already enqueued on the previous stream is waited for before the switch.
 * LIFETIME CONTRACT: the stream belongs to the caller and must outlive its use by the handle -- call
 * mw_ocean_use_own_stream (or mw_ocean_set_stream with another stream, or mw_ocean_destroy) BEFORE destroying it.  A
 * destroyed hipStream_t is a dangling pointer to HIP; the library tolerates the error codes the runtime returns for one
 * ("nothing pending") as a courtesy, but cannot make the use of a freed handle defined.                              */
mw_status mw_ocean_set_stream(mw_ocean* o, void* hip_stream);
mw_status mw_ocean_use_own_stream(mw_ocean* o);
void* mw_ocean_get_stream(mw_ocean* o);
mw_status mw_ocean_synchronize(mw_ocean* o);

/* choppiness / tDivision / mult may change between frames without regenerating the spectrum
 * (S/FFTMesh.cs:244-245 reads choppiness every frame; S/OceanRenderer.cs:96).                      */
mw_status mw_ocean_set_choppiness(mw_ocean* o, float choppiness);

/* Inject / read back verttilde and vertConj (S/FFTMesh.cs:35-36,114-116), N*N*2 floats each,
 * idx = i*N + j.  Injection is how a caller reproduces a Unity-generated spectrum exactly, and
 * how the parity tests feed identical inputs to the oracle and to the GPU.                         */
/* OceanRenderer semantics: the same pair is initialTexture.rg / .ba (F/InitialSpectrum.shader:53), M*M*2 floats
 * each with texel (px,py) at index py*M + px; setting it restarts the phase at 0.                              */
mw_status mw_ocean_set_spectrum(mw_ocean* o, const float* h0_xy, const float* h0conj_xy);
mw_status mw_ocean_get_spectrum(mw_ocean* o, float* h0_xy, float* h0conj_xy);

/* Regenerate the initial spectrum IN PLACE from new (length, wind, amplitude, seed); everything else of the handle stays.
 *  OceanRenderer semantics = the parameter-change branch of Update (S/OceanRenderer.cs:98-109): RenderInitial() draws
 *    initialTexture again (pass the handle's own seed: the reference keeps _RandomSeed1/2); the stateful phase textures are
 *    NOT touched, so the animation continues; dispersion and spectrum passes use the new length from the next frame on
 *    (:94-97), while the normal pass keeps the length it was created with -- the reference sets normalMat's _Length only in
 *    SetParams (:163) and never again.
 *  FFTMesh semantics = the spectrum fill of GenerateMesh (S/FFTMesh.cs:114-116) with a new seed (the `generate` tick
 *    draws fresh UnityEngine.Random values, :62-68); the timer is not touched (mw_ocean_reset_timer does that).  The grid
 *    must stay on the same evaluation path: MW_ESTATE if the new length flips unit_width == length / N.
 *  The call is synchronous and transactional: the new spectrum is generated into buffers of its own (the handle transiently holds
 *  two spectra: 2 x N^2 x 8 B more, 268 MB at 4096^2), the derived tables -- and for non-FFT grids the chirp tables of the new
 *  length -- are rebuilt, the stream is drained, and only then are the old buffers freed (hipFree: a device-wide synchronisation).  */
mw_status mw_ocean_reinit_spectrum(mw_ocean* o, float length, float wind_x, float wind_y, float amplitude, uint64_t seed);

/* Save / restore the animation state.  OceanRenderer: the current phase texture (F/Dispersion.shader:32-41), M*M floats,
 * texel (px,py) at py*M + px -- with initialTexture (mw_ocean_get_spectrum) this is everything a checkpoint needs.
 * FFTMesh: the timer (mw_ocean_timer / mw_ocean_set_timer); get/set_phase return MW_ESTATE there.                    */
mw_status mw_ocean_get_phase(mw_ocean* o, float* phase);
mw_status mw_ocean_set_phase(mw_ocean* o, const float* phase);
mw_status mw_ocean_set_timer(mw_ocean* o, float timer);
/* OceanRenderer: the length the normal pass divides by.  The reference sets normalMat._Length once in SetParams
 * (S/OceanRenderer.cs:163) and never again, so after mw_ocean_reinit_spectrum with a new length it differs from the
 * handle's length.  It is the third piece of an OceanRenderer checkpoint (with initialTexture and the phase): a fresh
 * handle created with the current length must be given the saved value.  FFTMesh handles return their length and
 * refuse the setter with MW_ESTATE.                                                                              */
float mw_ocean_normal_length(const mw_ocean* o);
mw_status mw_ocean_set_normal_length(mw_ocean* o, float normal_length);

/* GenerateMesh outputs (S/FFTMesh.cs:101-139, S/OceanRenderer.cs:172-207): rest vertices [N*N*3],
 * normals [N*N*3], uvs [N*N*2], triangle indices [(N-1)^2*6].  Any pointer may be NULL.            */
mw_status mw_ocean_rest_mesh(mw_ocean* o, float* vertices_xyz, float* normals_xyz, float* uvs_xy, int32_t* indices);
int64_t mw_ocean_index_count(const mw_ocean* o);
int32_t mw_ocean_grid_size(const mw_ocean* o); /* N of the synthesis grid (8*resolution in OceanRenderer mode) */

/* ---- per-frame: FFTMesh.EvaluateWaves(t)  (S/FFTMesh.cs:224-280) -------------------------------
 * host outputs: mesh.vertices [N*N*3], mesh.normals [N*N*3], mesh.colors [N*N*4].                  */
mw_status mw_ocean_evaluate(mw_ocean* o, float t, float* vertices_xyz, float* normals_xyz, float* colors_rgba);

/* FFTMesh.Update (S/FFTMesh.cs:60-73): timer += delta_time / tDivision; EvaluateWaves(timer).      */
mw_status mw_ocean_update(mw_ocean* o, float delta_time, float* vertices_xyz, float* normals_xyz, float* colors_rgba);
float mw_ocean_timer(const mw_ocean* o);
mw_status mw_ocean_reset_timer(mw_ocean* o); /* the `generate` tick of S/FFTMesh.cs:62-68 */

/* Throughput form: nsteps independent time-steps t[0..nsteps) of the same ocean in one enqueue
 * (FFTMesh semantics is a pure function of (h0, h0conj, t), S/FFTMesh.cs:178-190).  Outputs stay in
 * device memory: d_vertices [nsteps][N*N*3], d_normals [nsteps][N*N*3], d_white [nsteps][N*N] or
 * [nsteps][N*N*4] with MW_OUT_COLOR_RGBA.  t is a HOST array.  Asynchronous on the handle's stream. */
mw_status mw_ocean_evaluate_device(mw_ocean* o, const float* t, int32_t nsteps, void* d_vertices, void* d_normals,
                                   void* d_white, uint32_t flags);
int32_t mw_ocean_max_batch(const mw_ocean* o); /* largest nsteps one enqueue accepts */

/* ---- per-frame: OceanRenderer.GenerateTexture()  (S/OceanRenderer.cs:216-316) ------------------
 * advances the stateful phase by delta_time*mult and produces the four result textures, host side:
 * height [M*M] (= heightTexture.r), disp_xz [M*M*2] (= displacementTexture.rb),
 * normal_xyz [M*M*3] (= normalTexture.rgb), white [M*M] (= whiteTexture.r); M = 8*resolution,
 * texel (px,py) at index py*M + px.  Any pointer may be NULL.
 * Plan: these four planar textures need only Re h, Re / Im Dx and Re Dz, so the planar entry points run TWO complex transforms per frame
 * (height + i Dz share one, built from the Hermitian parts of the initial spectrum: csrc/ocean_renderer_kernels.h) where the shaders
 * run three; the RGBA forms below, whose channels include Im h and Im Dz, run three.  Both are within the stated float32 tolerance of
 * the reference's pipeline; one frame through the two forms may differ in the last bits.  A phase texture injected with
 * mw_ocean_set_phase that is not mirror-symmetric (every phase the library produced is) selects three transforms here too.        */
mw_status mw_ocean_generate_texture(mw_ocean* o, float delta_time, float* height, float* disp_xz, float* normal_xyz,
                                    float* white);
mw_status mw_ocean_generate_texture_device(mw_ocean* o, float delta_time, void* d_height, void* d_disp_xz,
                                           void* d_normal_xyz, void* d_white);

/* Throughput form: nframes CONSECUTIVE GenerateTexture() calls of one ocean in one enqueue.  Frame k advances the stateful phase by
 * delta_time[k]*mult on top of frame k-1 (F/Dispersion.shader:32-41, F/FFTCommon.cginc:101-104) -- a per-texel chain of one multiply,
 * one add and one fmod that the spectrum kernel walks in registers -- while the three transforms and the normal / whitecap passes of
 * different frames are independent and run as nframes-deep launches (S/OceanRenderer.cs:216-307 per frame).  Results, the phase
 * included, are bit-identical to nframes calls of mw_ocean_generate_texture_device.  delta_time is a HOST array; device destinations
 * are d_height [nframes][M*M], d_disp_xz [nframes][M*M*2], d_normal_xyz [nframes][M*M*3], d_white [nframes][M*M]; a NULL destination
 * keeps that texture's frames in the handle (mw_ocean_frame_textures).  Afterwards the handle's latest frame (mw_ocean_displace_mesh,
 * mw_ocean_get_phase) is frame nframes-1.  1 <= nframes <= mw_ocean_max_frames; single-ocean handles only (a handle of
 * mw_ocean_create_batch fills the device with tiles instead: MW_ESTATE).  Asynchronous on the handle's stream.                        */
mw_status mw_ocean_generate_texture_steps_device(mw_ocean* o, const float* delta_time, int32_t nframes, void* d_height,
                                                 void* d_disp_xz, void* d_normal_xyz, void* d_white);
/* The phase texture after nframes MORE GenerateTexture() calls with these delta times, without producing textures (the Dispersion pass alone,
 * F/Dispersion.shader:32-41; any nframes >= 0; delta_time is a HOST array): the handle then continues bit for bit like one that rendered
 * those frames.  How a rank of a multi-GPU job seeks to its own block of a frame sequence, and how a recorder skips frames.  Asynchronous.  */
mw_status mw_ocean_advance_phase(mw_ocean* o, const float* delta_time, int32_t nframes);
/* host form: the same nframes frames into host arrays [nframes][M*M*...] (any may be NULL); synchronous, PCIe-bound (28 B per texel and frame) */
mw_status mw_ocean_generate_texture_steps(mw_ocean* o, const float* delta_time, int32_t nframes, float* height, float* disp_xz,
                                          float* normal_xyz, float* white);
int32_t mw_ocean_max_frames(const mw_ocean* o); /* largest nframes one enqueue accepts (0: not an OceanRenderer handle) */
/* device pointers of frame `frame` of the LATEST steps call for the textures that call kept in the handle (NULL destination there);
 * a texture that went to a caller buffer reports NULL.  Valid until the next steps call of this handle.                              */
mw_status mw_ocean_frame_textures(mw_ocean* o, int32_t frame, void** d_height, void** d_disp_xz, void** d_normal_xyz, void** d_white);

/* ---- optional: page-lock caller arrays --------------------------------------------------------------------
 * The host-pointer entry points copy results into caller memory; into ordinary (pageable) arrays that copy runs at
 * ~9 GB/s and dominates the call.  A host that keeps its output arrays for many frames -- the
 * reference does: vertMeow/normals are allocated once (S/FFTMesh.cs:90-99) -- can register them once (in C#: after
 * GCHandle.Alloc(array, GCHandleType.Pinned)) and every later copy into them goes at PCIe rate.  Unregister before
 * the memory is freed or unpinned.                                                                               */
mw_status mw_host_register(void* ptr, size_t bytes);
mw_status mw_host_unregister(void* ptr);

/* ---- OceanRenderer semantics, consumer-side packing ---------------------------------------------------------
 * One GenerateTexture() delivered as the reference's four ARGBFloat render targets (S/OceanRenderer.cs:143-146,
 * bound to the ocean material at :310-313), [M*M*4] floats each, texel (px,py) at (py*M + px)*4:
 *   height_rgba = (Re h, Im h, Re h, Im h)      F/SpectrumHeight.shader:46 + F/Stockham.shader:56
 *   disp_rgba   = (Re Dx, Im Dx, Re Dz, Im Dz)  F/Spectrum.shader:50      + F/Stockham.shader:56
 *   normal_rgba = (n.x, n.y, n.z, 1)            F/OceanNormal.shader:55
 *   white_rgba  = (w, w, w, 1)                  F/WhiteCap.shader:44
 * Any destination may be NULL.  Host form synchronous, device form asynchronous on the handle's stream.          */
mw_status mw_ocean_generate_texture_rgba(mw_ocean* o, float delta_time, float* height_rgba, float* disp_rgba,
                                         float* normal_rgba, float* white_rgba);
mw_status mw_ocean_generate_texture_rgba_device(mw_ocean* o, float delta_time, void* d_height_rgba, void* d_disp_rgba,
                                                void* d_normal_rgba, void* d_white_rgba);
/* nframes consecutive frames (mw_ocean_generate_texture_steps_device) as the four ARGBFloat targets, [nframes][M*M*4] each.          */
mw_status mw_ocean_generate_texture_steps_rgba_device(mw_ocean* o, const float* delta_time, int32_t nframes, void* d_height_rgba,
                                                      void* d_disp_rgba, void* d_normal_rgba, void* d_white_rgba);
mw_status mw_ocean_generate_texture_steps_rgba(mw_ocean* o, const float* delta_time, int32_t nframes, float* height_rgba, float* disp_rgba,
                                               float* normal_rgba, float* white_rgba); /* host arrays [nframes][M*M*4], synchronous */
/* The ocean material's vertex stage on the resolution x resolution mesh of S/OceanRenderer.cs:172-207, sampling the
 * textures of the LATEST GenerateTexture() bilinearly at the vertex uv (tex2Dlod, clamp):
 *   vertex = rest + (_Anim.r, _Height.r, _Anim.b) / 8      W/TestOcean.shader:65-66, W/MistralWaterCommon.cginc:22-23
 *   normal = normalize(_Bump.rgb)                          W/TestOcean.shader:70
 *   color  = _White.r   (1 float per vertex)               W/TestOcean.shader:72, W/MistralWaterCommon.cginc:56
 * normals/colors may be NULL.  MW_ESTATE before the first GenerateTexture().                                      */
mw_status mw_ocean_displace_mesh(mw_ocean* o, float* vertices_xyz, float* normals_xyz, float* colors);
mw_status mw_ocean_displace_mesh_device(mw_ocean* o, void* d_vertices_xyz, void* d_normals_xyz, void* d_colors);

/* ---- independent tiles on several devices (BASELINE configs[2]) ------------------------------------
 * Tiles are independent units in both semantics: tile k is the ocean of `params` with seed params->seed + k on its own
 * device, compute stream and output buffers; there is no data-path collective.  FFTMesh tiles advance up to max_steps
 * independent time-steps per mw_tiles_evaluate; OceanRenderer tiles up to max_steps CONSECUTIVE frames per
 * mw_tiles_generate_texture_steps (mw_ocean_generate_texture_steps_device per tile; max_steps <= 32 in both semantics).
 * The only exchange is the optional gather of finished outputs to one root device over RCCL (xGMI), issued on per-device SIDE streams behind an event recorded on
 * the compute streams -- once per batch, never per step (29.4 MB per 1024^2 tile ~ 190 us on one 153 GB/s link).
 *   single process : mw_tiles_create -- one RCCL rank per distinct device (ncclCommInitAll, rccl.h:236); tiles that share
 *                    a device share its rank.  devices == NULL places tile k on device k % mw_device_count().
 *   one process per GPU (the torch.distributed.run launch of bench.py): rank 0 calls mw_comm_unique_id, the launcher
 *                    broadcasts the 128 bytes, every rank calls mw_tiles_create_rank (ncclCommInitRank, rccl.h:220) and
 *                    owns exactly one tile; mw_tiles_gather is then collective over the ranks.
 * RCCL is loaded with dlopen on first use (no link-time dependency); MW_EDEVICE when it is absent.                     */
typedef struct mw_tiles mw_tiles;
#define MW_COMM_ID_BYTES 128
mw_status mw_comm_unique_id(void* id_out); /* MW_COMM_ID_BYTES bytes (ncclGetUniqueId) */
mw_status mw_tiles_create(const mw_params* params, int32_t ntiles, const int32_t* devices, int32_t max_steps, mw_tiles** out);
mw_status mw_tiles_create_rank(const mw_params* params, int32_t device, int32_t max_steps, const void* comm_id, int32_t rank,
                               int32_t nranks, mw_tiles** out);
void mw_tiles_destroy(mw_tiles* t);
int32_t mw_tiles_count(const mw_tiles* t);       /* tiles in the whole job (= nranks in the per-process form) */
int32_t mw_tiles_local_count(const mw_tiles* t); /* tiles this process owns */
mw_ocean* mw_tiles_ocean(mw_tiles* t, int32_t local_k); /* borrowed handle of a local tile (set_spectrum, set_choppiness, ...) */
/* nsteps <= max_steps time-steps t[0..nsteps) on every local tile (mw_ocean_evaluate_device per tile, asynchronous) */
mw_status mw_tiles_evaluate(mw_tiles* t, const float* times, int32_t nsteps, uint32_t flags);
/* device pointers of local tile k's outputs of the LATEST mw_tiles_evaluate: [max_steps][N*N*3], [max_steps][N*N*3],
 * [max_steps][N*N*(1|4)].  A tile that is gathered owns two such sets used alternately (the gather sends straight from the set
 * the latest evaluate wrote while the next evaluate fills the other one -- no snapshot copy): ask again after every evaluate
 * once mw_tiles_gather is in use.  Without gathers the pointers never change.                                              */
mw_status mw_tiles_outputs(mw_tiles* t, int32_t local_k, void** d_vertices, void** d_normals, void** d_white);
/* OceanRenderer tiles: one GenerateTexture() (S/OceanRenderer.cs:216) on every local tile, asynchronous; the result textures
 * of local tile k stay in its handle: height [M*M], disp_xz [M*M*2], normal_xyz [M*M*3], white [M*M].                  */
mw_status mw_tiles_generate_texture(mw_tiles* t, float delta_time);
mw_status mw_tiles_textures(mw_tiles* t, int32_t local_k, void** d_height, void** d_disp_xz, void** d_normal_xyz, void** d_white);
/* OceanRenderer tiles created with max_steps > 1: nframes <= max_steps consecutive frames on every local tile in one enqueue per tile
 * (delta_time[nframes]: HOST array), asynchronous.  mw_tiles_frames: the frames of local tile k, height [max_steps][M*M], disp_xz
 * [max_steps][M*M*2], normal_xyz [max_steps][M*M*3], white [max_steps][M*M] (stable pointers; MW_ESTATE when max_steps is 1);
 * mw_tiles_textures stays the latest frame.  mw_tiles_gather(step) then collects FRAME `step` of the latest call.               */
mw_status mw_tiles_generate_texture_steps(mw_tiles* t, const float* delta_time, int32_t nframes);
mw_status mw_tiles_frames(mw_tiles* t, int32_t local_k, void** d_height, void** d_disp_xz, void** d_normal_xyz, void** d_white);
/* Collect step `step` of EVERY tile (OceanRenderer: frame `step` of the latest call; 0 with max_steps 1) on the device of tile `root` (global tile
 * index): asynchronous, on the side streams.  mw_tiles_gathered: the root's buffer, per tile [N*N*3 | N*N*3 | N*N*w] floats
 * (OceanRenderer: [M*M | M*M*2 | M*M*3 | M*M]); NULL on processes that do not own root.  All multi-device entry points put
 * the caller's current HIP device back before they return.                                                           */
mw_status mw_tiles_gather(mw_tiles* t, int32_t step, int32_t root);
mw_status mw_tiles_gathered(mw_tiles* t, void** d_gathered, int64_t* floats_per_tile);
mw_status mw_tiles_synchronize(mw_tiles* t); /* compute and side streams of every local tile */

/* ---- pond: Gerstner vertex displacement  (W/MistralWaterLib.cginc:71-99,154-180) ---------------
 * pos_xyz/out_xyz [nverts*3] world positions; waves [nwaves*3] = {dir.x, dir.y, speed}; amplitude is
 * the already x0.01-scaled _Amplitude (:172).  out = pos + offsets (:176).  Host pointers, synchronous. */
mw_status mw_gerstner_displace(const float* pos_xyz, int64_t nverts, const float* waves, int32_t nwaves,
                               float amplitude, float frequency, float steepness, float t, float* out_xyz,
                               int32_t device);
/* device-pointer form, asynchronous on hip_stream (NULL = default stream) */
mw_status mw_gerstner_displace_device(const void* d_pos_xyz, int64_t nverts, const float* waves, int32_t nwaves,
                                      float amplitude, float frequency, float steepness, float t, void* d_out_xyz,
                                      void* hip_stream);

/* many time values of one lattice in ONE launch (the positions are read once, the time part of every wave's phase is
 * joined by angle addition): t[nsteps] is a HOST array, d_out_xyz is [nsteps][nverts*3].  nwaves must be 4 or 8 and
 * nsteps * nwaves <= 256 (mw_gerstner_max_steps); otherwise MW_EINVAL.  Asynchronous on hip_stream.               */
mw_status mw_gerstner_displace_steps_device(const void* d_pos_xyz, int64_t nverts, const float* waves, int32_t nwaves,
                                            float amplitude, float frequency, float steepness, const float* t,
                                            int32_t nsteps, void* d_out_xyz, void* hip_stream);
int32_t mw_gerstner_max_steps(int32_t nwaves); /* 0 when this wave count has no batched kernel */

/* ---- pond: the material's whole vertex-stage Displacement()  (W/MistralWaterLib.cginc:154-180) with every
 * displacement mode of the shader library.  Fields are the material properties of W/MistralWaterProperty.cginc /
 * W/MistralWaterLib.cginc:53-66 under their own names; `amplitude` is the raw _Amplitude (the x0.01 of :134/:172 is
 * applied inside, as the shader does).  Object space = world space.                                             */
#define MW_POND_WAVE 0               /* Wave(), :127-152: y only, finite-difference normal with _Smoothing        */
#define MW_POND_GERSTNER 1           /* Gerstner(), :71-99: 4 waves = _WDirectionAB.xy/.zw, _WDirectionCD.xy/.zw  */
#define MW_POND_GERSTNER_LEVEL_ONE 2 /* GerstnerLevelOne(), :101-125: 5 built-in waves                           */
typedef struct mw_pond_params {
    int32_t mode;
    float amplitude, frequency, speed, steepness, smoothing; /* _Amplitude _Frequency _Speed _Steepness _Smoothing */
    float wspeed[4];                                         /* _WSpeed                                           */
    float dir_ab[4], dir_cd[4];                              /* _WDirectionAB, _WDirectionCD                      */
} mw_pond_params;
/* out_xyz = displaced vertex, out_normal_xyz (may be NULL) = v.normal as the shader leaves it.  Host pointers,
 * synchronous; t = _Time.y.                                                                                      */
mw_status mw_pond_displace(const mw_pond_params* p, const float* pos_xyz, int64_t nverts, float t, float* out_xyz,
                           float* out_normal_xyz, int32_t device);
/* device-pointer form, asynchronous on hip_stream (NULL = default stream) */
mw_status mw_pond_displace_device(const mw_pond_params* p, const void* d_pos_xyz, int64_t nverts, float t,
                                  void* d_out_xyz, void* d_out_normal_xyz, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* MISTRAL_WATER_H */
