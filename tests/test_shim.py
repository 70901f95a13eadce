"""The static forwarding shim (shim/libblst_eip2537.a): what the reference's untouched Rust
build script links (rust/build.rs:37-41).  No GPU needed: only host-side precompiles are called."""
import os
import subprocess

from conftest import ROOT

ABI13 = ["bls12_g1add", "bls12_g1mul", "bls12_g1multiexp", "bls12_g1multiexp_naive", "bls12_g1multiexp_bc",
         "bls12_g2add", "bls12_g2mul", "bls12_g2multiexp", "bls12_g2multiexp_naive", "bls12_g2multiexp_bc",
         "bls12_pairing", "bls12_map_fp_to_g1", "bls12_map_fp2_to_g2"]


def test_shim_archive_defines_abi_and_needs_only_libc(tmp_path, X):
    shim = os.path.join(ROOT, "shim")
    subprocess.check_call(["make", "-C", shim, "-s"])
    ar = os.path.join(shim, "libblst_eip2537.a")
    defined = {ln.split()[-1] for ln in subprocess.check_output(["nm", "--defined-only", ar], text=True).splitlines()
               if len(ln.split()) >= 3}
    for s in ABI13 + ["BLS12_MULTIEXP_DISCOUNT", "bls12_pairing_gas", "BLS12_G1MUL_GAS", "eip2537_hip_early_init"]:
        assert s in defined, s
    undefined = {ln.split()[-1] for ln in subprocess.check_output(["nm", "-u", ar], text=True).splitlines() if ln.strip() and ":" not in ln}
    allowed = {"dlopen", "dlsym", "dlerror", "getenv", "setenv", "fprintf", "__fprintf_chk", "stderr", "_GLOBAL_OFFSET_TABLE_"}
    assert undefined <= allowed, undefined - allowed      # libc only: no hip, no libstdc++
    # link a C client against the archive alone and run it against the engine .so, two threads
    import blst_eip2537_amd as pkg
    exe = str(tmp_path / "abi_client")
    subprocess.check_call(["gcc", "-O1", os.path.join(shim, "abi_client.c"), ar, "-lpthread", "-o", exe])
    env = dict(os.environ, EIP2537_HIP_LIB=pkg.lib_path())
    out = subprocess.run([exe], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert out.returncode == 0, out.stdout
    assert "0 failures" in out.stdout


def test_cgo_tree_builds_links_and_runs(tmp_path, X):
    """The Go drop-in tree of INTEGRATION.md (reference go/blst_eip2537.go:7-9 compiles every C / asm file of go/ by
    #include with -I../src -I../blst/bindings -I../blst/build -I../blst/src, -D__BLST_CGO__ -D__ADX__ -mno-avx and NO
    LDFLAGS; go/blst_c_files.c:7, go/blst_eip2537_c_files.c:7, go/blst_asm_file.S:1): the tree is built here from
    shim/eip2537_shim.c and include/eip2537.h, the three include-only translation units are written by this test (one line
    each, like the reference's), everything is compiled with the cgo flags, linked into a C main with no extra libraries
    and called through.  There is no Go toolchain here: gcc plays cgo."""
    import shutil
    tree = tmp_path / "tree"
    for d in ("src", "blst/bindings", "blst/build", "blst/src", "go"):
        (tree / d).mkdir(parents=True)
    shutil.copy(os.path.join(ROOT, "shim", "eip2537_shim.c"), tree / "src" / "eip2537.c")
    shutil.copy(os.path.join(ROOT, "include", "eip2537.h"), tree / "src" / "eip2537.h")
    (tree / "blst/bindings/blst.h").write_text("#ifndef __BLST_H__\n#define __BLST_H__\ntypedef unsigned char byte;\n#endif\n")
    (tree / "blst/src/server.c").write_text("")
    (tree / "blst/build/assembly.S").write_text("")
    units = {"blst_c_files.c": '#include "server.c"\n', "blst_eip2537_c_files.c": '#include "eip2537.c"\n',
             "blst_asm_file.S": '#include "assembly.S"\n'}
    go = tree / "go"
    cflags = ["-D__BLST_CGO__", "-D__ADX__", "-mno-avx", "-I../src", "-I../blst/bindings", "-I../blst/build", "-I../blst/src",
              "-g", "-O2", "-fPIC"]
    objs = []
    for name, text in units.items():
        (go / name).write_text(text)
        obj = name.rsplit(".", 1)[0] + ".o"
        subprocess.check_call(["gcc"] + cflags + ["-c", name, "-o", obj], cwd=go)
        objs.append(str(go / obj))
    # the caller: the C client of the shim test, compiled against the tree's own header like a cgo preamble would
    main_c = go / "main.c"
    main_c.write_text(open(os.path.join(ROOT, "shim", "abi_client.c")).read().replace('"../include/eip2537.h"', '"eip2537.h"'))
    subprocess.check_call(["gcc"] + cflags + ["-c", "main.c", "-o", "main.o"], cwd=go)
    exe = str(tmp_path / "cgo_client")
    subprocess.check_call(["gcc", str(go / "main.o")] + objs + ["-lpthread", "-o", exe])      # -lpthread: what the Go runtime brings anyway; no -ldl, no hip
    und = {ln.split()[-1].split("@")[0] for ln in subprocess.check_output(["nm", "-u", exe], text=True).splitlines() if ln.strip()}
    assert not any(u.startswith(("hip", "bls12_", "eip2537_")) for u in und), und
    import blst_eip2537_amd as pkg
    out = subprocess.run([exe], env=dict(os.environ, EIP2537_HIP_LIB=pkg.lib_path()), stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                         text=True, timeout=120)
    assert out.returncode == 0 and "0 failures" in out.stdout, out.stdout
    # zero-length call with a dangling pointer and the wrong-length rule, through the tree (go/blst_eip2537.go:45-47 never sends
    # an empty slice, Rust does)
    probe = go / "probe.c"
    probe.write_text('#include <stdio.h>\n#include "eip2537.h"\nint main(void) { byte out[128]; '
                     'int a = bls12_g1multiexp(out, (byte *)8, 0), b = bls12_g1add(out, (const byte *)8, 255); '
                     'printf("%d %d\\n", a, b); return !(a == EIP2537_INVALID_LENGTH && b == EIP2537_INVALID_LENGTH); }\n')
    subprocess.check_call(["gcc"] + cflags + ["probe.c"] + objs + ["-o", "probe"], cwd=go)
    out = subprocess.run([str(go / "probe")], env=dict(os.environ, EIP2537_HIP_LIB=pkg.lib_path()), stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, text=True, timeout=120)
    assert out.returncode == 0, out.stdout
