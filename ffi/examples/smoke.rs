//! One worker on GPU 0 through the safe layer of plonk_hip.rs, the way `impl plonk_slave::Server for PlonkImpl` (worker.rs:125-439)
//! would drive it: init with a two-point SRS, one var_msm, one local 2-D transform (fft_init / fft1 / fft2_prepare / fft2).
//! UNCOMPILED in this repository's image (no Rust toolchain) — it documents the call sequence; tests/host_cpp/host_check.cpp is the
//! same sequence as compiled C++ and runs in the GPU test suite.
use plonk_hip::*;

fn main() -> Result<(), PlonkError> {
    let mut st = GpuState::new(0, PLONK_BN254)?;
    // BN254 G1 generator (1, 2) in arkworks' in-memory GroupAffine layout: x, y Montgomery limbs, infinity flag, padding (72 bytes)
    let one_mont: [u64; 4] /* 1 in Fq */ = [0xd35d438dc58f0d9d, 0x0a78eb28f5c70b3d, 0x666ea36f7879462c, 0x0e0a77c19a07df2f];
    let two_mont: [u64; 4] = [0xa6ba871b8b1e1b3a, 0x14f1d651eb8e167b, 0xccdd46def0f28c58, 0x1c14ef83340fbe5e];
    let mut bases = Vec::new();
    for _ in 0..2 {
        for w in one_mont.iter().chain(two_mont.iter()) {
            bases.extend_from_slice(&w.to_le_bytes());
        }
        bases.extend_from_slice(&[0u8; 8]); // infinity = false + padding
    }
    st.init(&bases, 72, 4, 32)?;
    // 3*G + 4*G = 7*G
    let scalars: [u64; 8] = [3, 0, 0, 0, 4, 0, 0, 0];
    let mut jac = [0u64; 12];
    st.var_msm(0, 2, &scalars, &mut jac)?;
    println!("var_msm -> X = {:016x?}", &jac[0..4]);
    // a 4-point transform as the reference distributes it: r = c = 2, one worker owning both rows and both columns
    let wl = [plonk_fft_workload { row_start: 0, row_end: 2, col_start: 0, col_end: 2 }];
    st.fft_init(1, &wl, 0, false, false, false)?;
    let fr_one: [u64; 4] = [0xac96341c4ffffffb, 0x36fc76959f60cd29, 0x666ea36f7879462e, 0x0e0a77c19a07df2f]; // 1 in Fr, Montgomery (utils.rs:27-43)
    let row0: [u64; 8] = [fr_one[0], fr_one[1], fr_one[2], fr_one[3], 0, 0, 0, 0];
    let row1: [u64; 8] = [0; 8];
    st.fft1(1, 0, &row0)?;
    st.fft1(1, 1, &row1)?;
    st.fft2_prepare(1)?;
    let mut cols = [0u64; 16];
    st.fft2(1, &mut cols)?;
    // the transform of (1, 0, 0, 0) is (1, 1, 1, 1)
    assert!(cols.chunks(4).all(|c| c == &fr_one[..]));
    println!("smoke ok");
    Ok(())
}
