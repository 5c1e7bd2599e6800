"""Latent cache between the inversion and composition stages.

Reference wire format kept for drop-in: one ``{dir}/ddim_latents_{t}.pt`` per timestep, ``torch.save`` of a
``[1,4,F,h,w]`` fp16 tensor holding the latent AT noise level t (write ``pipeline_i2vgen_xl.py:1988-1993``, read
``utils.py:31-36``; composition re-reads three of them per step, ``pipeline_i2vgen_xl.py:1637,1648-1650,1670``).

The reference saves synchronously inside the inversion loop and ``torch.load``s + H2D-copies inside the
composition loop.  Here every latent stays resident in HBM (524 KB each; a 500-step inversion is 262 MB of a 288 GB
device) and files are written by a background thread from a pinned host copy, so neither loop ever waits on disk.
"""
import os
import queue
import threading

import torch


def latent_file(directory, t):
    return os.path.join(directory, f"ddim_latents_{int(t)}.pt")


class LatentCache:
    def __init__(self, device, write_files=True):
        self.device = torch.device(device)
        self.write_files = write_files
        self._mem = {}
        self._q = queue.Queue()
        self._worker = None
        self._errors = []

    @staticmethod
    def _key(directory, t):
        return (os.path.abspath(directory), int(t))

    def _ensure_worker(self):
        if self._worker is None or not self._worker.is_alive():
            self._worker = threading.Thread(target=self._drain, daemon=True)
            self._worker.start()

    def _drain(self):
        while True:
            item = self._q.get()
            try:
                if item is None:
                    return
                path, host, event = item
                event.synchronize()  # the pinned copy has landed
                os.makedirs(os.path.dirname(path), exist_ok=True)
                tmp = path + ".tmp"
                torch.save(host.clone(), tmp)
                os.replace(tmp, path)
            except Exception as e:  # surfaced by flush()
                self._errors.append(e)
            finally:
                self._q.task_done()

    def put(self, directory, t, latents):
        """latents: device tensor [1,4,F,h,w] fp16 (not modified afterwards by the caller)"""
        if directory is None:
            return
        self._mem[self._key(directory, t)] = latents
        if not self.write_files:
            return
        if latents.is_cuda:
            host = torch.empty(latents.shape, dtype=latents.dtype, device="cpu", pin_memory=True)
            host.copy_(latents, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        else:
            host = latents

            class _Done:
                @staticmethod
                def synchronize():
                    pass
            ev = _Done()
        self._ensure_worker()
        self._q.put((latent_file(directory, t), host, ev))

    def get(self, directory, t):
        """device tensor [1,4,F,h,w] fp16 for noise level t: from HBM if this process produced it, else from disk"""
        key = self._key(directory, t)
        hit = self._mem.get(key)
        if hit is None:
            path = latent_file(directory, t)
            assert os.path.exists(path), f"Missing latents at t {t} path {path}"
            hit = torch.load(path, map_location="cpu").to(self.device, torch.float16).contiguous()
            self._mem[key] = hit
        return hit

    def flush(self):
        self._q.join()
        if self._errors:
            raise RuntimeError(f"latent cache writer failed: {self._errors[0]!r}")

    def clear(self):
        self.flush()
        self._mem.clear()
