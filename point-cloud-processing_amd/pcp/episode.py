"""Streaming an episode through the GPU path: file decode on host threads, pinned staging, H2D on a
copy stream overlapped with the kernels of the previous chunk, (T,K,6) DP3 rows back in pinned
memory.  Replaces the per-frame loops XPC:96-125 + DP3:68-90 + CHK:134-144 for one trajectory
without the intermediate PLY files."""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import _ffi, ops


class EpisodeRunner:
    def __init__(self, H, W, K, params, device="cuda", chunk=64, dist_mode=None, Hc=None, Wc=None, depth_mm=False,
                 stage_threads=8, upload_piece=25, rgb_on_host=True, profile=False):
        """rgb_on_host (default): the rgb images never leave host memory.  align_colors' mask needs rgb.shape only
        (XPC:72-75) and colours are wanted for the K sampled points (DP3:48), so the device returns each sampled
        point's colour-image pixel (pcp_sampled_color_pixels) and the rows' colour half is gathered from the host
        images: per 640x480 frame 0.6 MB (u16 depth) or 1.2 MB (f32) cross PCIe instead of 1.5 / 2.2 MB.  False: rgb
        is uploaded and gathered on the device (pcp_gather_rgb_pixels) — same rows, bit for bit.
        profile: accumulate a host / PCIe / kernel time breakdown in self.stats (adds timing events)."""
        if not torch.cuda.is_available():
            raise RuntimeError("pcp needs a ROCm device (MI355X); there is no CPU fallback")
        self.H, self.W, self.K, self.params, self.chunk = H, W, K, params, int(chunk)
        self.depth_mm = bool(depth_mm)          # chunks carry raw uint16 millimetres: half the upload, /1000 on the device
        self.Hc, self.Wc = Hc or H, Wc or W
        self.dev = torch.device(device)
        if self.dev.index is None:
            self.dev = torch.device("cuda", torch.cuda.current_device())
        if dist_mode is None:                   # the loop this replaces samples with use_cuda=True (DP3:47)
            dist_mode = ops.dist_mode_for(True)
        self.dist_mode = int(dist_mode)
        self.pipe = ops.FramePipeline(self.chunk, H, W, K, params, self.dev, dist_mode=self.dist_mode)
        self.copy_stream = torch.cuda.Stream(device=self.dev)
        mk = lambda shape, dt: torch.empty(shape, dtype=dt).pin_memory()
        ddt = torch.uint16 if self.depth_mm else torch.float32
        self.h_depth = [mk((self.chunk, H, W), ddt) for _ in range(2)]
        self.rgb_on_host = bool(rgb_on_host)
        self.d_depth = [torch.empty((self.chunk, H, W), dtype=ddt, device=self.dev) for _ in range(2)]
        self.d_depth_m = torch.empty((self.chunk, H, W), dtype=torch.float32, device=self.dev) if self.depth_mm else None
        if self.rgb_on_host:
            self.h_rgb = self.d_rgb = None
            self.h_xyz = [mk((self.chunk, K, 3), torch.float32) for _ in range(3)]
            self.h_cpix = [mk((self.chunk, K), torch.int32) for _ in range(3)]
            self.h_rows = None
        else:
            self.h_rgb = [mk((self.chunk, self.Hc, self.Wc, 3), torch.uint8) for _ in range(2)]
            self.d_rgb = [torch.empty((self.chunk, self.Hc, self.Wc, 3), dtype=torch.uint8, device=self.dev) for _ in range(2)]
        self.profile = bool(profile)
        self.stats = dict(frames=0, chunks=0, stage_s=0.0, wait_s=0.0, host_rows_s=0.0, h2d_ms=0.0, kernels_ms=0.0, d2h_ms=0.0)
        self._ev = [[torch.cuda.Event(enable_timing=True) for _ in range(6)] for _ in range(3)] if self.profile else None
        self.uploaded = [torch.cuda.Event() for _ in range(2)]
        self.consumed = [torch.cuda.Event() for _ in range(2)]
        if not self.rgb_on_host:
            self.h_rows = [mk((self.chunk, K, 6), torch.float32) for _ in range(3)]
        # input buffers exist twice (slot = chunk parity), result buffers and timing events three times (chunk % 3): the
        # collector may still be reading chunk i-1's results while chunk i+1 is uploaded into chunk i-1's input slot
        self.h_off = [mk((self.chunk + 1,), torch.int64) for _ in range(3)]
        self.h_flags = [mk((2,), torch.int32) for _ in range(3)]      # stage-1 capacity / FPS bounds flags of the chunk
        self.done = [torch.cuda.Event() for _ in range(3)]
        self._collector = ThreadPoolExecutor(max_workers=1)
        self.stage_threads = max(1, int(stage_threads))
        self.upload_piece = int(upload_piece)                # frames per staging/DMA piece
        self._pool = ThreadPoolExecutor(max_workers=self.stage_threads) if self.stage_threads > 1 else None
        self._gpool = ThreadPoolExecutor(max_workers=self.stage_threads) if self.stage_threads > 1 else None   # the collector's

    def close(self):
        """Stop the staging threads (also called when the runner is garbage-collected)."""
        if self._pool is not None:
            self._pool.shutdown(wait=True)
            self._pool = None
        if getattr(self, "_gpool", None) is not None:
            self._gpool.shutdown(wait=True)
            self._gpool = None
        if getattr(self, "_collector", None) is not None:
            self._collector.shutdown(wait=True)
            self._collector = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _upload(self, slot, rslot, depth, rgb):
        """Stage + upload one chunk in pieces.  Every piece's pageable → pinned copy is ONE task of the staging pool (numpy
        releases the GIL inside copyto), all pieces are submitted at once, and the main thread enqueues a piece's H2D copy as
        soon as that piece is staged: the copies run back to back on the copy stream while later pieces are still being
        staged, so staging and the PCIe transfer overlap instead of adding up."""
        import time
        n = depth.shape[0]
        self.uploaded[slot].synchronize()                    # the slot's pinned staging buffer: its last H2D has finished
        if n < self.chunk:                                   # a short tail chunk: invalid depth → empty clouds
            self.h_depth[slot][n:].zero_()
        piece = max(1, self.upload_piece)
        with torch.cuda.stream(self.copy_stream):
            self.copy_stream.wait_event(self.consumed[slot])
            if self.profile:
                self._ev[rslot][0].record(self.copy_stream)
        hd = self.h_depth[slot].numpy()
        hc = self.h_rgb[slot].numpy() if not self.rgb_on_host else None

        def stage(a, b):
            np.copyto(hd[a:b], depth[a:b])
            if hc is not None:
                np.copyto(hc[a:b], rgb[a:b])

        cuts = list(range(0, n, piece)) + [n] if n > 0 else [0, 0]
        t0 = time.perf_counter()
        if self._pool is not None:
            futs = [self._pool.submit(stage, a, b) for a, b in zip(cuts[:-1], cuts[1:])]
        else:
            futs = None
        for k, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
            if futs is not None:
                futs[k].result()
            elif b > a:
                stage(a, b)
            last = k == len(cuts) - 2
            hi = self.chunk if last else b                   # the last piece carries the zeroed tail too
            with torch.cuda.stream(self.copy_stream):
                self.d_depth[slot][a:hi].copy_(self.h_depth[slot][a:hi], non_blocking=True)
                if not self.rgb_on_host:
                    self.d_rgb[slot][a:hi].copy_(self.h_rgb[slot][a:hi], non_blocking=True)
                if last:
                    if self.profile:
                        self._ev[rslot][1].record(self.copy_stream)
                    self.uploaded[slot].record(self.copy_stream)
        self.stats["stage_s"] += time.perf_counter() - t0

    def run_chunks(self, chunks):
        """chunks: iterable of (depth (n,H,W) f32, rgb (n,Hc,Wc,3) u8) with n <= chunk.
        Returns (T,K,6) float32 DP3 rows and the (T,) cropped point counts.
        Order per chunk: enqueue its kernels + D2H, THEN decode/stage/upload the next chunk on the host
        and the copy stream while the GPU works, then collect the rows."""
        with torch.cuda.device(self.dev):                    # events, streams and kernels all belong to self.dev
            return self._run_chunks(chunks)

    def _host_rows(self, slot, n, rgb):
        """(n,K,6) DP3 rows from the chunk's xyz + colour pixels and the HOST rgb images: colours[mask] = rgb[v_idx,
        u_idx] / 255 (XPC:77), zero where masked (XPC:76), times 255 again in the row (CHK:141) = the u8 value."""
        out = np.empty((n, self.K, 6), np.float32)
        xyz, cp, hw = self.h_xyz[slot].numpy(), self.h_cpix[slot].numpy(), self.Hc * self.Wc

        def part(ab):                                   # frames [a, b): numpy's take releases the GIL
            a, b = ab
            out[a:b, :, :3] = xyz[a:b]
            c = cp[a:b]
            flat = np.ascontiguousarray(rgb[a:b]).reshape((b - a) * hw, 3).view("V3").reshape(-1)   # pixels as 3-byte items
            rows = np.maximum(c, 0).astype(np.int64)
            rows += (np.arange(b - a, dtype=np.int64) * hw)[:, None]
            got = np.take(flat, rows.reshape(-1)).view(np.uint8).reshape(b - a, self.K, 3)   # one gather of (b-a)*K pixels
            np.multiply(got, (c >= 0)[:, :, None], out=out[a:b, :, 3:], casting="unsafe")

        parts = min(self.stage_threads, n) if self._gpool is not None else 1
        if parts <= 1:
            part((0, n))
        else:
            cuts = [n * i // parts for i in range(parts + 1)]
            list(self._gpool.map(part, zip(cuts[:-1], cuts[1:])))
        return out

    def _collect(self, rslot, n, rgb, chunk_no):
        """Collector thread: wait for chunk `chunk_no`'s results (pinned buffers of result slot `rslot`) and turn them into
        rows while the GPU works on the next chunk."""
        import time
        t0 = time.perf_counter()
        self.done[rslot].synchronize()
        t1 = time.perf_counter()
        fl = self.h_flags[rslot]
        if int(fl[0]) or int(fl[1]):
            raise _ffi.PcpError(-3, "stage-1 row capacity exceeded" if int(fl[0]) else
                                "a cloud exceeded max_pts / total_pts in pcp_fps_batched")
        rows = self._host_rows(rslot, n, rgb) if self.rgb_on_host else self.h_rows[rslot][:n].numpy().copy()
        counts = np.diff(self.h_off[rslot].numpy())[:n]
        st = self.stats
        st["wait_s"] += t1 - t0; st["host_rows_s"] += time.perf_counter() - t1
        st["frames"] += n; st["chunks"] += 1
        if self.profile:                                    # the chunk's events have all completed by now
            ev = self._ev[rslot]
            st["h2d_ms"] += ev[0].elapsed_time(ev[1]); st["kernels_ms"] += ev[2].elapsed_time(ev[3])
            st["d2h_ms"] += ev[3].elapsed_time(ev[4])
        return rows, counts

    def _run_chunks(self, chunks):
        """Three things overlap: the GPU runs chunk i, the main thread stages + uploads chunk i+1 (copy stream), the
        collector thread turns chunk i-1's results into rows."""
        futures = []
        main = torch.cuda.current_stream(self.dev)
        for s in range(2):
            self.consumed[s].record(main)
        it = iter(chunks)
        nxt = next(it, None)
        i = 0
        if nxt is not None:
            self._upload(0, 0, *nxt)
        while nxt is not None:
            cur_n, slot, rs = nxt[0].shape[0], i & 1, i % 3
            main.wait_event(self.uploaded[slot])
            ev = self._ev[rs] if self.profile else None
            if ev:
                ev[2].record(main)
            dsrc = ops.depth_mm_to_m(self.d_depth[slot], self.d_depth_m) if self.depth_mm else self.d_depth[slot]
            if self.rgb_on_host:
                xyz, cpix, idx, off = self.pipe.run_host_rgb(dsrc, self.Hc, self.Wc)
                self.consumed[slot].record(main)
                if ev:
                    ev[3].record(main)
                self.h_xyz[rs].copy_(xyz, non_blocking=True)
                self.h_cpix[rs].copy_(cpix, non_blocking=True)
            else:
                xyz, col, idx, off = self.pipe.run(dsrc, self.d_rgb[slot])
                packed = ops.pack_dp3(xyz, col)
                self.consumed[slot].record(main)
                if ev:
                    ev[3].record(main)
                self.h_rows[rs].copy_(packed, non_blocking=True)
            self.h_off[rs].copy_(off, non_blocking=True)
            self.h_flags[rs][0:1].copy_(self.pipe.stage1.overflow, non_blocking=True)
            self.h_flags[rs][1:2].copy_(self.pipe.fps.overflow, non_blocking=True)
            if ev:
                ev[4].record(main)
            self.done[rs].record(main)
            futures.append(self._collector.submit(self._collect, rs, cur_n, nxt[1] if self.rgb_on_host else None, i))
            nxt = next(it, None)
            i += 1
            if nxt is not None:
                if i >= 3:
                    futures[i - 3].result()                      # result slot i % 3 (buffers, events) is free again
                self._upload(i & 1, i % 3, *nxt)
        if not futures:
            return np.zeros((0, self.K, 6), np.float32), np.zeros((0,), np.int64)
        got = [f.result() for f in futures]
        return np.concatenate([g[0] for g in got]), np.concatenate([g[1] for g in got])

    def run_arrays(self, depth, rgb):
        T = depth.shape[0]
        ddt = np.uint16 if self.depth_mm else np.float32
        gen = ((np.ascontiguousarray(depth[i:i + self.chunk], dtype=ddt),
                np.ascontiguousarray(rgb[i:i + self.chunk], dtype=np.uint8)) for i in range(0, T, self.chunk))
        return self.run_chunks(gen)

    def run_tree(self, extracted_root, workers=8):
        """<extracted_root>/camera/{rgb/rgb_%05d.png, depth/depth_%05d.npy} (XTP:234-246 layout)."""
        from PIL import Image
        rgb_dir = os.path.join(extracted_root, "camera", "rgb")
        dep_dir = os.path.join(extracted_root, "camera", "depth")
        n_files = len([f for f in os.listdir(rgb_dir) if f.endswith(".png")])
        ids = [i for i in range(n_files) if os.path.exists(os.path.join(rgb_dir, f"rgb_{i:05d}.png"))
               and os.path.exists(os.path.join(dep_dir, f"depth_{i:05d}.npy"))]              # XPC:100-101

        def load(i):
            d = np.load(os.path.join(dep_dir, f"depth_{i:05d}.npy")).astype(np.float32)
            c = np.asarray(Image.open(os.path.join(rgb_dir, f"rgb_{i:05d}.png")).convert("RGB"))
            return d, c

        def gen():
            with ThreadPoolExecutor(max_workers=workers) as pool:
                for i in range(0, len(ids), self.chunk):
                    got = list(pool.map(load, ids[i:i + self.chunk]))
                    yield np.stack([g[0] for g in got]), np.stack([g[1] for g in got])

        rows, counts = self.run_chunks(gen())
        return rows, counts, ids
