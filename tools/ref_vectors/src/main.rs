//! Emits reference vectors for the batched authenticated-share path of ark-mpc as JSON on stdout.
//! Everything is computed by ark-mpc / arkworks 0.4 themselves; this file only chooses inputs and prints.
//! See README.md for how the vectors are consumed (tests/test_ref_vectors.py).

use ark_bn254::G1Projective as Bn254;
use ark_curve25519::EdwardsProjective as Ed25519;
use ark_ec::CurveGroup;
use ark_mpc::{
    algebra::{AuthenticatedScalarResult, CurvePoint, Scalar},
    network::{NetworkOutbound, NetworkPayload},
    test_helpers::execute_mock_mpc,
    PARTY0,
};
use num_bigint::BigUint;
use serde_json::{json, Value};
use sha3::{Digest, Sha3_256};

fn hex(bytes: &[u8]) -> String {
    bytes.iter().map(|b| format!("{b:02x}")).collect()
}

fn dec<C: CurveGroup>(s: &Scalar<C>) -> String {
    s.to_biguint().to_string()
}

fn scalar_from_dec<C: CurveGroup>(d: &str) -> Scalar<C> {
    Scalar::from_biguint(&d.parse::<BigUint>().unwrap())
}

/// Edge and generic values: 0, 1, 2, r-1, r-2, 2^64, 2^128 + 12345, 2^192 - 1, 2^250 + 7 and a few fixed words (all reduced mod r)
fn test_scalars<C: CurveGroup>() -> Vec<Scalar<C>> {
    let minus_one = Scalar::<C>::from(0u8) - Scalar::<C>::from(1u8);
    let mut v = vec![
        Scalar::from(0u8),
        Scalar::from(1u8),
        Scalar::from(2u8),
        minus_one,
        minus_one - Scalar::from(1u8),
        Scalar::from_biguint(&(BigUint::from(1u8) << 64)),
        Scalar::from_biguint(&((BigUint::from(1u8) << 128) + BigUint::from(12345u32))),
        Scalar::from_biguint(&((BigUint::from(1u8) << 192) - BigUint::from(1u8))),
        Scalar::from_biguint(&((BigUint::from(1u8) << 250) + BigUint::from(7u8))),
    ];
    // a deterministic tail: x_{k+1} = x_k^2 + 0x9E3779B97F4A7C15
    let mut x = Scalar::<C>::from(0xA11CE001u64);
    for _ in 0..8 {
        x = x * x + Scalar::from(0x9E3779B97F4A7C15u64);
        v.push(x);
    }
    v
}

fn scalar_vectors<C: CurveGroup>(curve: &str) -> Vec<Value> {
    test_scalars::<C>()
        .iter()
        .map(|s| {
            json!({
                "curve": curve,
                "value_dec": dec(s),
                "to_bytes_be_hex": hex(&s.to_bytes_be()),                      // scalar.rs:118-127
                "serde_json": String::from_utf8(serde_json::to_vec(s).unwrap()).unwrap(), // scalar.rs:186-192: 32 LE bytes as numbers
            })
        })
        .collect()
}

fn point_vectors<C: CurveGroup>(curve: &str) -> Vec<Value> {
    let g = CurvePoint::<C>::generator();
    test_scalars::<C>()
        .iter()
        .map(|k| {
            let p = g * *k;                                                   // curve.rs:403-409
            json!({
                "curve": curve,
                "scalar_dec": dec(k),
                "to_bytes_hex": hex(&p.to_bytes()),                           // curve.rs:103-108 (serialize_compressed)
                "neg_to_bytes_hex": hex(&(-p).to_bytes()),
                "double_to_bytes_hex": hex(&(p + p).to_bytes()),
            })
        })
        .collect()
}

fn wire_vectors<C: CurveGroup>(curve: &str) -> Vec<Value> {
    let scalars = test_scalars::<C>();
    let g = CurvePoint::<C>::generator();
    let points: Vec<CurvePoint<C>> = scalars.iter().take(5).map(|k| g * *k).collect();
    let mut out = Vec::new();
    for (result_id, n) in [(6usize, 0usize), (7, 1), (1234567, scalars.len())] {
        let msg = NetworkOutbound::<C> { result_id, payload: NetworkPayload::ScalarBatch(scalars[..n].to_vec()) };
        out.push(json!({
            "curve": curve, "variant": "ScalarBatch", "result_id": result_id,
            "values_dec": scalars[..n].iter().map(dec).collect::<Vec<_>>(),
            "json": String::from_utf8(serde_json::to_vec(&msg).unwrap()).unwrap(),   // network.rs:33-60; quic.rs:303-306 prefixes a u64 LE length
        }));
    }
    let msg = NetworkOutbound::<C> { result_id: 99, payload: NetworkPayload::PointBatch(points.clone()) };
    out.push(json!({
        "curve": curve, "variant": "PointBatch", "result_id": 99,
        "scalars_dec": scalars.iter().take(5).map(dec).collect::<Vec<_>>(),
        "points_to_bytes_hex": points.iter().map(|p| hex(&p.to_bytes())).collect::<Vec<_>>(),
        "json": String::from_utf8(serde_json::to_vec(&msg).unwrap()).unwrap(),
    }));
    out
}

/// commitment.rs:71-86 with the public methods it calls: SHA3-256 over to_bytes_be(v_0) || ... || to_bytes_be(blinder),
/// reduced with from_be_bytes_mod_order (HashCommitment itself is pub(crate))
fn commitment_vectors<C: CurveGroup>(curve: &str) -> Vec<Value> {
    let scalars = test_scalars::<C>();
    let mut out = Vec::new();
    for n in [0usize, 1, 4, scalars.len()] {
        let blinder = scalars[scalars.len() - 1 - (n % 3)];
        let mut hasher = Sha3_256::new();
        for v in &scalars[..n] {
            hasher.update(v.to_bytes_be());
        }
        hasher.update(blinder.to_bytes_be());
        let digest = hasher.finalize();
        let commitment = Scalar::<C>::from_be_bytes_mod_order(&digest);
        out.push(json!({
            "curve": curve,
            "values_dec": scalars[..n].iter().map(dec).collect::<Vec<_>>(),
            "blinder_dec": dec(&blinder),
            "sha3_256_hex": hex(&digest),
            "commitment_dec": dec(&commitment),
        }));
    }
    out
}

/// AuthenticatedScalarResult::batch_mul (authenticated_scalar.rs:848-879) under execute_mock_mpc with PartyIDBeaverSource: each party's
/// local ScalarShare of the shared inputs and of every product, and the authenticated opening (bn254: the reference's TestCurve)
async fn batch_mul_vectors() -> Value {
    let xs: Vec<String> = test_scalars::<Bn254>().iter().map(dec).collect();
    let ys: Vec<String> = test_scalars::<Bn254>().iter().rev().map(dec).collect();
    let (xs2, ys2) = (xs.clone(), ys.clone());
    let (p0, p1) = execute_mock_mpc(move |fabric| {
        let (xs, ys) = (xs2.clone(), ys2.clone());
        async move {
            let x: Vec<Scalar<Bn254>> = xs.iter().map(|d| scalar_from_dec(d)).collect();
            let y: Vec<Scalar<Bn254>> = ys.iter().map(|d| scalar_from_dec(d)).collect();
            let a = fabric.batch_share_scalar(x, PARTY0);                      // fabric.rs:578-600
            let b = fabric.batch_share_scalar(y, PARTY0);
            let prod = AuthenticatedScalarResult::batch_mul(&a, &b);
            let mut rows = Vec::new();
            for (i, p) in prod.iter().enumerate() {
                let (sa, sb, sp) = (a[i].clone().await, b[i].clone().await, p.clone().await);   // this party's ScalarShare values
                rows.push(json!({
                    "x_share": [dec(&sa.share()), dec(&sa.mac())],
                    "y_share": [dec(&sb.share()), dec(&sb.mac())],
                    "product_share": [dec(&sp.share()), dec(&sp.mac())],
                }));
            }
            let opened = AuthenticatedScalarResult::open_authenticated_batch(&prod);
            let mut vals = Vec::new();
            for o in opened {
                vals.push(dec(&o.await.expect("MAC check")));
            }
            json!({ "shares": rows, "opened_dec": vals })
        }
    })
    .await;
    json!({ "curve": "bn254", "source": "PartyIDBeaverSource", "x_dec": xs, "y_dec": ys, "party0": p0, "party1": p1 })
}

#[tokio::main]
async fn main() {
    let mut scalars = scalar_vectors::<Bn254>("bn254");
    scalars.extend(scalar_vectors::<Ed25519>("curve25519"));
    let mut points = point_vectors::<Bn254>("bn254");
    points.extend(point_vectors::<Ed25519>("curve25519"));
    let mut wire = wire_vectors::<Bn254>("bn254");
    wire.extend(wire_vectors::<Ed25519>("curve25519"));
    let mut commitments = commitment_vectors::<Bn254>("bn254");
    commitments.extend(commitment_vectors::<Ed25519>("curve25519"));
    let doc = json!({
        "generator": "tools/ref_vectors: renegade-fi/ark-mpc (online-phase) over ark-* 0.4, sha3 0.10, serde_json 1",
        "scalars": scalars,
        "points": points,
        "wire": wire,
        "commitments": commitments,
        "batch_mul": batch_mul_vectors().await,
    });
    println!("{}", serde_json::to_string_pretty(&doc).unwrap());
}
