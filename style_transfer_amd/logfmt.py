"""Output side of the boundary: the reference's log grammar and the get_i_iter.py extraction.

Every record the reference emits through print_and_log (siamese_contrastive.py:1590-1592) appears in
the log file as `INFO:siamese_contrastive.py:print_and_log:1592 - <msg>` (logging format at
:1604-1605); outputs/get_i_iter.py keys on `Sentence Pair.`, `iteration.N,` and `Change.N,`.
"""
LOG_PREFIX = "INFO:siamese_contrastive.py:print_and_log:1592 - "


def headline(sent: str, note_token="*"):
    """utils.py:8-10."""
    return note_token * 20 + sent + note_token * 20


class LogWriter:
    """print_and_log: message to stdout (optional) and to the log file with the reference's prefix."""

    def __init__(self, path=None, echo=True):
        self.f = open(path, "w") if path else None
        self.echo = echo
        self.lines = []

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
