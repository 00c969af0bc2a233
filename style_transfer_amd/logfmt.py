"""Output side of the boundary: the reference's log grammar and the get_i_iter.py extraction.

Every record the reference emits through print_and_log (siamese_contrastive.py:1590-1592) appears in
the log file as `INFO:siamese_contrastive.py:print_and_log:1592 - <msg>` (logging format at
:1604-1605); outputs/get_i_iter.py keys on `Sentence Pair.`, `iteration.N,` and `Change.N,`.
"""
LOG_PREFIX = "INFO:siamese_contrastive.py:print_and_log:1592 - "


def headline(sent: str, note_token="*"):
    """utils.py:8-10."""
    return note_token * 20 + sent + note_token * 20


class VocabBytes:
    """id_to_word as one UTF-8 byte string + offsets: the form st_format_attack_log reads"""

    def __init__(self, id_to_word):
        import numpy as np
        enc = [w.encode("utf-8") for w in id_to_word]
        self.bytes = b"".join(enc)
        self.off = np.zeros(len(enc) + 1, np.int64)
        np.cumsum([len(b) for b in enc], out=self.off[1:])
        self.n = len(enc)


def format_attack_block(vocab: VocabBytes, data, rec, first_index=0, eos=3, pad=0, unk=1, prefix=LOG_PREFIX):
    """The log block of len(data) sentences as bytes (numpy uint8), formatted by the library's host function
    st_format_attack_log (csrc/st_logfmt.cpp) from a CPU LeanRecord (attack.LeanRecord) and the origins of `data`
    (an AttackSet covering exactly these sentences).  Same bytes as attack.write_attack_log through a LogWriter."""
    import numpy as np
    from . import _lib
    L = _lib.lib()
    S = len(data)
    assert len(rec) == S
    T = int(rec.init_ids.shape[1])
    I = int(rec.changed.shape[1])
    npy = lambda t, dt: None if t is None else np.ascontiguousarray(t.numpy(), dt)
    n_iter, init, ch, rows = npy(rec.n_iter, np.int32), npy(rec.init_ids, np.int32), npy(rec.changed, np.uint8), npy(rec.rows, np.int32)
    assert int(ch.sum()) == rows.shape[0], "changed flags and changed rows disagree"
    like, unlike = npy(rec.like_ids, np.int32), npy(rec.unlike_ids, np.int32)
    kp = 0 if like is None else like.shape[1]
    kn = 0 if unlike is None else unlike.shape[1]
    src = data.origin_r
    src_tok, src_off = np.ascontiguousarray(src.tokens, np.int32), np.ascontiguousarray(src.off, np.int64)
    ptr = lambda a: None if a is None else a.ctypes.data
    args = (vocab.bytes, vocab.off.ctypes.data, vocab.n, eos, pad, unk, prefix.encode(), S, T, I, int(first_index),
            ptr(src_tok), ptr(src_off), ptr(init), ptr(like), kp, ptr(unlike), kn, ptr(n_iter), ptr(ch), ptr(rows))
    need = L.st_format_attack_log(*args, None, 0)
    if need < 0:
        raise ValueError("st_format_attack_log: bad argument")
    out = np.empty(need, np.uint8)
    got = L.st_format_attack_log(*args, out.ctypes.data, need)
    assert got == need
    return out


def format_attack_blocks(vocab: VocabBytes, data, rec, first_index=0, threads=8, min_chunk=4096, **kw):
    """format_attack_block over a large CPU record (rank 0 after the gather of a multi-GPU run): the sentence range is
    cut into chunks formatted on `threads` host threads (st_format_attack_log is a C function: ctypes drops the GIL)
    and yielded in order."""
    import torch
    from concurrent.futures import ThreadPoolExecutor
    S = len(data)
    n = max(1, min(threads * 4, S // max(1, min_chunk)))
    if n == 1:
        yield format_attack_block(vocab, data, rec, first_index, **kw)
        return
    row_start = torch.zeros(S + 1, dtype=torch.int64)
    row_start[1:] = torch.cumsum(rec.changed.sum(1, dtype=torch.int64), 0)
    cuts = [S * i // n for i in range(n + 1)]
    with ThreadPoolExecutor(threads) as ex:
        futs = [ex.submit(format_attack_block, vocab, data.span(cuts[i], cuts[i + 1]), rec.span(cuts[i], cuts[i + 1], row_start),
                          first_index + cuts[i], **kw) for i in range(n)]
        for f in futs:
            yield f.result()


class LogWriter:
    """print_and_log: message to stdout (optional) and to the log file with the reference's prefix."""

    def __init__(self, path=None, echo=True):
        self.f = open(path, "w") if path else None
        self.echo = echo
        self.lines = []

    def write_block(self, block):
        """a pre-formatted block of records (format_attack_block): bytes with prefix and newlines"""
        if self.echo or not self.f:
            raw = block.tobytes() if hasattr(block, "tobytes") else bytes(block)
            msgs = [l[len(LOG_PREFIX):] for l in raw.decode("utf-8").split("\n")[:-1]]
            if self.echo:
                print("\n".join(msgs))
            if not self.f:
                self.lines.extend(msgs)
        if self.f:
            self.f.flush()
            # (a contiguous byte array goes to the file as it is: tobytes() copied 265 MB per 98 304-sentence batch)
            flags = getattr(block, "flags", None)
            if flags is not None and flags["C_CONTIGUOUS"]:
                self.f.buffer.write(memoryview(block))
            else:
                self.f.buffer.write(block.tobytes() if hasattr(block, "tobytes") else bytes(block))
            self.f.buffer.flush()

    def __call__(self, msg: str):
        if self.echo:
            print(msg)
        if self.f:
            self.f.write(LOG_PREFIX + msg + "\n")
        else:
            self.lines.append(msg)

    def close(self):
        if self.f:
            self.f.close()


def extract_out(lines, pos_c=0, neg_c=0, change=None, it=None, limit=1000):
    """outputs/get_i_iter.py:50-141 (default and --it modes) on an iterable of log lines.
    Returns the list of output lines (each keeps its trailing newline, as the script writes them)."""
    import re
    res = {}
    sent_id = -1
    for i in lines:
        if not i.endswith("\n"):
            i = i + "\n"
        if re.search("Sentence Pair.", i):
            sent_id += 1
            continue
        for tag in ("iteration.", "Change."):
            m = re.search(tag, i)
            if m:
                first = m.start() + len(tag)
                it_id = int(i[first:].split(",")[0])
                res.setdefault("%s%d" % (tag, it_id), {})[sent_id] = i[first + len(str(it_id)) + 2:]
    out = []
    for i in range(limit):
        if it is not None:
            out.append(res["iteration.%d" % it][i])
            continue
        c = change if change is not None else (pos_c if i < 500 else neg_c)
        sent = None
        for cur in range(c, -1, -1):
            name = "Change.%d" % cur
            if name in res and i in res[name]:
                sent = res[name][i]
                break
        if sent is None:
            sent = res["iteration.0"][i]
        out.append(sent)
    return out
