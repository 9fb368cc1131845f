"""Test-side restatement of moshi-server's BatchedAsr host logic (srv/batched_asr.rs:69-116, 527-718, 796-808) in
plain Python over any engine-like object (encode_step / step_tokens / reset_batch_idx / poll_msgs), plus the glue that
lets the C worker run on the CPU oracle.  Both are driven by the same script of events; what each channel receives
must be identical."""
import ctypes as C
import heapq

import numpy as np

FRAME = 1920


class RefWorker:
    def __init__(self, eng, batch_size, delay, n_heads, detok=None):
        self.eng, self.B, self.delay, self.nh, self.detok = eng, batch_size, delay, n_heads, detok
        self.ch = [None] * batch_size
        self.active, self.free = [], list(range(batch_size))
        self.markers, self.seq, self.next_id = [], 0, 1
        self.enc_step, self.model_step = 0, 0
        self.batch_pcm = np.zeros((batch_size, FRAME), dtype=np.float32)

    def open(self):
        if not self.free:
            raise RuntimeError("Server at capacity - no free channels available")
        slot = self.free.pop(0)
        self.ch[slot] = dict(id=self.next_id, inq=[{"type": "Init"}], out=[], data=np.zeros(0, np.float32), closed=False)
        self.next_id += 1
        self.active.append(slot)
        return slot

    def send(self, slot, msg):
        self.ch[slot]["inq"].append(msg)

    def close_channel(self, slot):
        self.ch[slot]["closed"] = True

    def recv(self, slot):
        c = self.ch[slot]
        out, c["out"] = c["out"], []
        return out

    def _extend(self, c, pcm, out_pcm):  # Channel::extend_data
        if pcm.size == 0 and c["data"].size < FRAME:
            return False
        c["data"] = np.concatenate([c["data"], pcm])
        if c["data"].size >= FRAME:
            out_pcm[:] = c["data"][:FRAME]
            c["data"] = c["data"][FRAME:]
            return True
        return False

    def _emit(self, slot, msg, ref_id):  # Channel::send
        c = self.ch[slot]
        if c is None or c["id"] != ref_id:
            return
        if c["closed"]:
            self.ch[slot] = None
            return
        c["out"].append(msg)

    def step(self):
        mask = np.zeros(self.B, dtype=np.uint8)
        ids = [0] * self.B
        resets, new_markers = [], []
        for bid in self.active:
            c = self.ch[bid]
            self.batch_pcm[bid] = 0
            if c is None:
                continue
            ids[bid] = c["id"]
            if c["closed"]:
                continue
            m = False
            while True:
                if not c["inq"]:
                    if self._extend(c, np.zeros(0, np.float32), self.batch_pcm[bid]):
                        m = True
                    break
                msg = c["inq"].pop(0)
                if msg["type"] == "Init":
                    c["out"].append({"type": "Ready"})
                    resets.append(bid)
                elif msg["type"] == "Marker":
                    new_markers.append((self.enc_step + self.delay + c["data"].size // FRAME, self.seq, bid, c["id"], msg["id"]))
                    self.seq += 1
                elif msg["type"] == "Audio":
                    if self._extend(c, np.asarray(msg["pcm"], dtype=np.float32), self.batch_pcm[bid]):
                        m = True
            mask[bid] = 1 if m else 0
        for bid in list(self.active):
            c = self.ch[bid]
            if c is None or c["closed"]:
                self.ch[bid] = None
                self.active.remove(bid)
                self.free.append(bid)
        if not mask.any() and not resets and not new_markers:
            return False
        self.enc_step += 1
        codes = self.eng.encode_step(self.batch_pcm, mask)
        for bid in resets:
            self.eng.reset_batch_idx(bid)
        text, prs = self.eng.step_tokens(np.where(mask[:, None].astype(bool), codes, 0), mask)
        self.model_step += 1
        for mk in new_markers:
            heapq.heappush(self.markers, mk)
        for am in self.eng.poll_msgs():
            if am[0] == "Word":
                _, bid, t, toks = am
                txt = self.detok(toks) if self.detok else " ".join(str(x) for x in toks)
                self._emit(bid, {"type": "Word", "text": txt, "start_time": t}, ids[bid])
            elif am[0] == "EndWord":
                _, bid, t = am
                self._emit(bid, {"type": "EndWord", "stop_time": t}, ids[bid])
            else:
                for b in range(self.B):
                    if mask[b] and self.ch[b] is not None:
                        self._emit(b, {"type": "Step", "step_idx": am[1], "prs": [float(prs[h][b]) for h in range(self.nh)],
                                       "buffered_pcm": int(self.ch[b]["data"].size)}, ids[b])
        while self.markers and self.markers[0][0] <= self.model_step:
            _, _, bid, cid, mid = heapq.heappop(self.markers)
            self._emit(bid, {"type": "Marker", "id": mid}, cid)
        return True


def oracle_backend(dsm, ora, cfg, B):
    """A dsm_worker_backend whose four calls go to an OracleAsr (tests only: the product backend is the HIP engine)."""
    state = {"codes": np.zeros((B, cfg.mimi.quantizer_n_q), dtype=np.uint32)}

    def encode(_s, pcm, mask):
        p = np.ctypeslib.as_array(pcm, shape=(B, FRAME)).copy()
        m = np.ctypeslib.as_array(mask, shape=(B,)).copy()
        state["codes"] = np.where(m[:, None].astype(bool), ora.encode_step(p, m, side=0), 0).astype(np.uint32)
        return 0

    def reset(_s, slot):
        ora.reset_batch_idx(slot)
        return 0

    def step(_s, mask, text, prs):
        m = np.ctypeslib.as_array(mask, shape=(B,)).copy()
        t, p = ora.step_tokens(state["codes"], m)
        np.ctypeslib.as_array(text, shape=(B,))[:] = t
        if cfg.extra_heads_num:
            np.ctypeslib.as_array(prs, shape=(cfg.extra_heads_num, B))[:] = p
        return 0

    def poll(_s, msgs, cap, toks, tcap):
        return ora.L.orc_asr_poll_msgs(ora.h, msgs, cap, C.cast(toks, C.c_void_p), tcap)

    be = dsm.WorkerBackend()
    be.batch_size, be.asr_delay_in_tokens, be.extra_heads_num = B, cfg.asr_delay_in_tokens, cfg.extra_heads_num
    cbs = (dsm.BE_ENCODE(encode), dsm.BE_RESET(reset), dsm.BE_STEP(step), dsm.BE_POLL(poll))
    be.encode_step, be.reset_slot, be.step_tokens, be.poll_msgs = cbs
    return be, cbs


def script(B, n_steps, seed=3):
    """Events per worker step: channels open late, audio arrives in ragged chunks (less and more than a frame), markers,
    pings, undecodable frames, a client that leaves and whose slot is taken over by a newcomer, and a full house."""
    rng = np.random.default_rng(seed)
    ev = []
    for s in range(n_steps):
        e = []
        if s == 0:
            e += [("open",)] * (B - 1)
        if s == 2:
            e += [("open",), ("open_expect_full",)]
        if s == 9:
            e += [("close", 1)]
        if s == 11:
            e += [("open",)]  # takes over slot 1
        for slot in range(B):
            if rng.random() < 0.85:
                n = int(rng.integers(200, 4500))
                e.append(("audio", slot, (0.1 * rng.standard_normal(n)).astype(np.float32)))
            if rng.random() < 0.15:
                e.append(("marker", slot, int(rng.integers(-5, 1000))))
            if rng.random() < 0.1:
                e.append(("ping", slot))
            if rng.random() < 0.05:
                e.append(("garbage", slot))
        e.append(("step",))
        ev.append(e)
    return ev
