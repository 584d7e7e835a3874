"""The background compiler's job directories (rusterix_amd/csrc/rxr_jit.hip): they live under ONE private parent of the user, and
the clean-up of stale ones walks by file descriptor without ever following a symbolic link (round-2 advisor finding: the first
version swept /tmp/rxr_jit_* by name and would have emptied whatever directory a link of that name pointed at)."""
import ctypes as C
import os
import stat
import subprocess
import sys
import time

import rusterix_amd


def _lib():
    lib = C.CDLL(rusterix_amd.lib_paths()["rxr"])
    lib.rxr_debug_jit_sweep.argtypes = [C.c_char_p]
    lib.rxr_debug_jit_sweep.restype = None
    lib.rxr_debug_jit_job_parent.argtypes = [C.c_char_p, C.c_uint32]
    return lib


def _ns():
    return os.stat("/proc/self/ns/pid").st_ino


def _dead_pid():
    p = subprocess.Popen([sys.executable, "-c", "pass"])
    p.wait()
    return p.pid


def _job(parent, name, owner=None, age=0.0):
    d = parent / name
    d.mkdir()
    (d / "set.h").write_text("x")
    sub = d / "comgr-123"
    sub.mkdir()
    (sub / "tmp.o").write_text("y")
    if owner is not None:
        (d / "owner").write_text(owner)
    if age:
        t = time.time() - age
        os.utime(d, (t, t))
    return d


def test_sweep_removes_only_stale_directories_of_this_user_and_follows_no_link(tmp_path):
    lib = _lib()
    parent = tmp_path / "parent"
    parent.mkdir(mode=0o700)
    victim = tmp_path / "victim"
    victim.mkdir()
    (victim / "precious.txt").write_text("keep me")
    (victim / "sub").mkdir()
    (victim / "sub" / "deep.txt").write_text("keep me too")

    dead = _job(parent, "job_dead", f"{_dead_pid()} {_ns()}\n")                    # owner gone, our namespace: stale
    alive = _job(parent, "job_alive", f"{os.getpid()} {_ns()}\n")                  # owner alive: kept
    other_ns_new = _job(parent, "job_otherns", f"{_dead_pid()} {_ns() + 1}\n")     # another PID namespace, recent: kept (the pid says nothing)
    other_ns_old = _job(parent, "job_otherns_old", f"{_dead_pid()} {_ns() + 1}\n", age=7200)  # ... an hour old: stale
    no_owner_new = _job(parent, "job_noowner")                                     # half written, recent: kept
    no_owner_old = _job(parent, "job_noowner_old", age=7200)                       # stale
    os.symlink(victim, parent / "job_link")                                        # a link named like a job: never examined
    os.utime(parent / "job_link", (0, 0), follow_symlinks=False)
    (dead / "evil").symlink_to(victim)                                             # a link INSIDE a stale job: removed as a link
    (dead / "evil_file").symlink_to(victim / "precious.txt")
    not_a_job = parent / "other_thing"
    not_a_job.mkdir()

    lib.rxr_debug_jit_sweep(str(parent).encode())

    assert not dead.exists() and not other_ns_old.exists() and not no_owner_old.exists()
    assert alive.exists() and (alive / "comgr-123" / "tmp.o").exists()
    assert other_ns_new.exists() and no_owner_new.exists() and not_a_job.exists()
    assert (parent / "job_link").is_symlink()
    assert (victim / "precious.txt").read_text() == "keep me" and (victim / "sub" / "deep.txt").read_text() == "keep me too"


def test_job_parent_is_a_private_directory(tmp_path, monkeypatch):
    lib = _lib()
    buf = C.create_string_buffer(4096)
    # XDG_RUNTIME_DIR when it is a private directory of ours ...
    run = tmp_path / "run"
    run.mkdir(mode=0o700)
    monkeypatch.setenv("XDG_RUNTIME_DIR", str(run))
    assert lib.rxr_debug_jit_job_parent(buf, len(buf)) == 0
    p = buf.value.decode()
    assert p == str(run / "rxr_jit")
    st = os.lstat(p)
    assert stat.S_ISDIR(st.st_mode) and st.st_uid == os.geteuid() and (st.st_mode & 0o077) == 0
    # ... refused when somebody could have planted the parent: group / other bits set
    os.chmod(p, 0o777)
    assert lib.rxr_debug_jit_job_parent(buf, len(buf)) != 0 and b"not a private directory" in buf.value
    os.chmod(p, 0o700)
    # ... or when it is a symbolic link
    os.rmdir(p)
    elsewhere = tmp_path / "elsewhere"
    elsewhere.mkdir(mode=0o700)
    os.symlink(elsewhere, p)
    assert lib.rxr_debug_jit_job_parent(buf, len(buf)) != 0
    # a world-writable XDG_RUNTIME_DIR is not trusted: /tmp/rxr_jit-<uid>
    loose = tmp_path / "loose"
    loose.mkdir()
    os.chmod(loose, 0o777)
    monkeypatch.setenv("XDG_RUNTIME_DIR", str(loose))
    assert lib.rxr_debug_jit_job_parent(buf, len(buf)) == 0
    assert buf.value.decode() == f"/tmp/rxr_jit-{os.geteuid()}"
